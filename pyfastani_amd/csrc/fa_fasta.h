// Host ingest: FASTA records straight from a memory-mapped file.
//
// Record semantics follow the reference's own parser (src/pyfastani/_fasta.pyx:41-103): the file yields records only
// if its first line starts with '>'; the identifier is the whole header line without '>' and the newline; sequence
// lines are concatenated with their one trailing '\n' removed and ASCII letters upper-cased (copy_upper, toupper
// semantics); a header that does not end in a newline within the reference's 2048-byte line buffer is an error
// (BufferError there).  One deliberate difference: the reference reads sequence lines in 2047-byte pieces and would
// take a piece that happens to start with '>' for a header; here only real line starts count.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fa_common.h"

namespace fa {

struct FastaFile {
  int fd = -1;
  const char *data = nullptr;
  size_t size = 0;
  size_t pos = 0;          // start of the next header line
  bool started = false;    // the first line began with '>'
  std::string id;
  std::vector<uint8_t> seq;

  FastaFile() = default;
  FastaFile(const FastaFile &) = delete;
  FastaFile &operator=(const FastaFile &) = delete;
  ~FastaFile() { close(); }

  void open(const char *path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) throw Error(FA_ERR_IO, std::string(path) + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { int e = errno; close(); throw Error(FA_ERR_IO, std::string(path) + ": " + strerror(e)); }
    if (S_ISDIR(st.st_mode)) { close(); throw Error(FA_ERR_IO, std::string(path) + ": is a directory"); }
    size = (size_t)st.st_size;
    if (size > 0) {
      void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p == MAP_FAILED) { int e = errno; close(); throw Error(FA_ERR_IO, std::string(path) + ": mmap: " + strerror(e)); }
      data = (const char *)p;
      (void)madvise(p, size, MADV_SEQUENTIAL);
    }
    pos = 0;
    started = size > 0 && data[0] == '>';
  }

  void close() {
    if (data) munmap((void *)data, size);
    data = nullptr; size = 0;
    if (fd >= 0) ::close(fd);
    fd = -1;
  }

  struct Span { size_t header, header_end, body, body_end; };   // header line [header, header_end) incl. '\n'; body up to the next header

  // boundaries of the next record, without copying anything
  bool next_span(Span &sp) {
    if (!started || pos >= size || data[pos] != '>') return false;
    const char *nl = (const char *)memchr(data + pos, '\n', size - pos);
    const size_t line_len = nl ? (size_t)(nl - (data + pos)) + 1 : size - pos;
    // fgets(line, 2048): at most 2047 characters, and the reference insists that the last one is the newline
    if (!nl || line_len > 2047) throw Error(FA_ERR_BUFFER, "FASTA identifier too large for the line buffer");
    sp.header = pos; sp.header_end = pos + line_len; sp.body = sp.header_end;
    size_t p = sp.body;
    while (p < size && data[p] != '>') {
      const char *e = (const char *)memchr(data + p, '\n', size - p);
      p = e ? (size_t)(e - data) + 1 : size;
    }
    sp.body_end = p;
    pos = p;
    return true;
  }

  static void fill_id(const char *data, const Span &sp, std::string &id) { id.assign(data + sp.header + 1, sp.header_end - sp.header - 2); }

  // sequence bytes of a record: lines joined, ASCII letters upper-cased
  static void fill_seq(const char *data, const Span &sp, std::vector<uint8_t> &seq) {
    seq.resize(sp.body_end - sp.body);
    size_t n = 0, p = sp.body;
    while (p < sp.body_end) {
      const char *e = (const char *)memchr(data + p, '\n', sp.body_end - p);
      const size_t end = e ? (size_t)(e - data) : sp.body_end;
      for (size_t i = p; i < end; i++) {
        uint8_t c = (uint8_t)data[i];
        seq[n++] = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
      }
      p = e ? end + 1 : end;
    }
    seq.resize(n);
  }

  bool next() {
    Span sp;
    if (!next_span(sp)) return false;
    fill_id(data, sp, id);
    fill_seq(data, sp, seq);
    return true;
  }
};

// every record of a file: boundaries found serially (memchr speed), bodies joined and upper-cased by a pool of threads
inline void read_fasta_records(const char *path, std::vector<std::vector<uint8_t>> &seqs, int threads) {
  FastaFile f;
  f.open(path);
  std::vector<FastaFile::Span> spans;
  FastaFile::Span sp;
  while (f.next_span(sp)) spans.push_back(sp);
  seqs.assign(spans.size(), {});
  const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), spans.size()));
  if (nt <= 1) {
    for (size_t i = 0; i < spans.size(); i++) FastaFile::fill_seq(f.data, spans[i], seqs[i]);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
      pool.emplace_back([&, t] { for (size_t i = (size_t)t; i < spans.size(); i += (size_t)nt) FastaFile::fill_seq(f.data, spans[i], seqs[i]); });
    for (auto &th : pool) th.join();
  }
}

}  // namespace fa
