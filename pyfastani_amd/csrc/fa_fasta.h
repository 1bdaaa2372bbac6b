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
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "fa_error.h"
#include "fa_host.h"   // HostPool

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
    // the next header is a '>' at a line start; sequence lines hold no '>', so this is normally one memchr per record
    size_t p = sp.body;
    while (p < size) {
      const char *e = (const char *)memchr(data + p, '>', size - p);
      if (!e) { p = size; break; }
      p = (size_t)(e - data);
      if (p == sp.body || data[p - 1] == '\n') break;
      p++;
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

// Every record of a file.  Boundaries are found serially (memchr speed); the bodies are then cut into pieces of about
// 256 KiB that start at line starts, every piece counts the bytes it will produce, and after a prefix sum all pieces
// are joined + upper-cased in parallel straight into place -- a genome that is ONE long record still uses every thread.
struct FastaSeq {
  std::unique_ptr<uint8_t[]> data;   // not value-initialised: the pieces below write every byte
  size_t size = 0;
};

inline void read_fasta_records(const char *path, std::vector<FastaSeq> &seqs) {
  FastaFile f;
  f.open(path);
  std::vector<FastaFile::Span> spans;
  FastaFile::Span sp;
  while (f.next_span(sp)) spans.push_back(sp);
  seqs.clear();
  seqs.resize(spans.size());
  struct Piece { size_t rec, lo, hi, out, count; };
  std::vector<Piece> pieces;
  const size_t PIECE = 256 * 1024;
  for (size_t r = 0; r < spans.size(); r++) {
    size_t p = spans[r].body;
    while (p < spans[r].body_end) {
      size_t q = std::min(spans[r].body_end, p + PIECE);
      if (q < spans[r].body_end) {                       // extend to the end of the line
        const char *e = (const char *)memchr(f.data + q, '\n', spans[r].body_end - q);
        q = e ? (size_t)(e - f.data) + 1 : spans[r].body_end;
      }
      pieces.push_back({r, p, q, 0, 0});
      p = q;
    }
  }
  const char *data = f.data;
  HostPool::get().parallel_for(pieces.size(), [&](size_t i) {
    Piece &pc = pieces[i];
    size_t nl = 0;
    for (const char *c = data + pc.lo, *end = data + pc.hi; (c = (const char *)memchr(c, '\n', (size_t)(end - c))) != nullptr; c++) nl++;
    pc.count = (pc.hi - pc.lo) - nl;
  });
  std::vector<size_t> total(spans.size(), 0);
  for (auto &pc : pieces) { pc.out = total[pc.rec]; total[pc.rec] += pc.count; }
  for (size_t r = 0; r < spans.size(); r++) { seqs[r].size = total[r]; seqs[r].data.reset(new uint8_t[std::max<size_t>(total[r], 1)]); }
  HostPool::get().parallel_for(pieces.size(), [&](size_t i) {
    const Piece &pc = pieces[i];
    uint8_t *dst = seqs[pc.rec].data.get() + pc.out;
    size_t p = pc.lo;
    while (p < pc.hi) {
      const char *e = (const char *)memchr(data + p, '\n', pc.hi - p);
      const size_t end = e ? (size_t)(e - data) : pc.hi;
      for (size_t k = p; k < end; k++) {
        const uint8_t c = (uint8_t)data[k];
        *dst++ = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
      }
      p = e ? end + 1 : end;
    }
  });
}

}  // namespace fa
