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

#include <atomic>
#include <cerrno>
#include <chrono>
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
  bool borrowed = false;   // `data` belongs to the caller (attach) or to `piped`
  std::vector<char> piped; // the bytes of a path that is not a regular file (a FIFO, a process substitution), read to their end
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
    if (!S_ISREG(st.st_mode)) {
      // no size to map: take the bytes until end of file (st_size is 0 for a pipe -- it used to come back as an empty genome)
      size_t got = 0;
      piped.resize((size_t)1 << 20);
      for (;;) {
        if (got == piped.size()) piped.resize(piped.size() * 2);
        const ssize_t r = ::read(fd, piped.data() + got, piped.size() - got);
        if (r < 0) { if (errno == EINTR) continue; const int e = errno; close(); throw Error(FA_ERR_IO, std::string(path) + ": read: " + strerror(e)); }
        if (r == 0) break;
        got += (size_t)r;
      }
      ::close(fd); fd = -1;
      data = piped.data(); size = got; borrowed = true; pos = 0;
      started = size > 0 && data[0] == '>';
      return;
    }
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

  // the same reader over bytes the caller holds (read_fasta_packed: a file read into a reusable per-thread buffer)
  void attach(const char *bytes, size_t n) {
    close();
    data = bytes; size = n; pos = 0; borrowed = true;
    started = size > 0 && data[0] == '>';
  }

  void close() {
    if (data && !borrowed) munmap((void *)data, size);
    borrowed = false;
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


// ----------------------------------------------------------------------------------------------------------
// One FASTA file, read and packed by ONE thread in a single sweep (round 5).
//
// read_fasta_records above spreads ONE file over the pool (two parallel sweeps + the packer's third): right for one big
// file, wrong for a thousand genomes, which were read strictly one after another at 2.9 GB/s on 256 host threads.  Here a file
// is one task: the bytes are read() into a per-thread buffer that is reused from file to file (no mmap / munmap per file,
// no page faults on an intermediate: a thousand 5 MB files had gone through 5 GB of freshly mapped join buffers), the records
// are found at memchr speed, and the sequence lines go STRAIGHT into 2-bit words -- 32 bases per AVX2 step, the same step
// finds the newline -- without the joined upper-case copy in between.  Every record starts on a 64-base boundary of the
// file's own word array, exactly as it will in a sequence store, so placing a record (or a prefix of it: a query batch
// holds whole fragments only) into a store is a copy of words; exceptions (bytes outside ACGT/acgt, upper-cased) carry
// file-relative base offsets.  Record semantics as FastaFile (src/pyfastani/_fasta.pyx:41-103).  Protein files keep their
// upper-cased bytes.
// ----------------------------------------------------------------------------------------------------------
// One anonymous mapping handed out in slices: the packed words of MANY files.  A fresh 1.3 MB block per file had made the first
// call of a process three to four times slower than its steady state (1 000 mmaps + 3 x 10^5 page faults of 4 KB under 25
// threads, every one a visit to the address space's lock: per-task sums of 3.8 ms read + 2.0 ms pack per 5 MB file against 2 ms
// all in all once the allocator was warm); the arena is one mapping, 2 MB aligned so that transparent huge pages serve it.
struct HostArena {
  char *raw = nullptr, *base = nullptr;
  size_t raw_bytes = 0, bytes = 0;
  std::atomic<size_t> used{0};
  explicit HostArena(size_t n) {
    const size_t huge = (size_t)2 << 20;
    bytes = (n + huge - 1) / huge * huge;
    raw_bytes = bytes + huge;
    void *p = mmap(nullptr, raw_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) throw Error(FA_ERR_NOMEM, "mmap of the packing arena failed");
    raw = (char *)p;
    base = (char *)(((uintptr_t)raw + huge - 1) / huge * huge);
    (void)madvise(base, bytes, MADV_HUGEPAGE);
  }
  HostArena(const HostArena &) = delete;
  HostArena &operator=(const HostArena &) = delete;
  ~HostArena() { if (raw) munmap(raw, raw_bytes); }
  // n bytes on a 64-byte boundary, or nullptr when the arena is used up (the caller allocates for itself then)
  void *take(size_t n) {
    n = (n + 63) / 64 * 64;
    const size_t at = used.fetch_add(n);
    return at + n <= bytes ? base + at : nullptr;
  }
};

struct PackedFasta {
  bool protein = false;
  uint32_t *words = nullptr;           // nucleotide: 16 bases per word (A=0 C=1 G=2 T=3; anything else 0 + an exception)
  uint8_t *bytes = nullptr;            // protein: upper-cased residues
  std::unique_ptr<uint32_t[]> own_words;   // (the arrays above live here, or in a slice of `arena`)
  std::unique_ptr<uint8_t[]> own_bytes;
  std::shared_ptr<HostArena> arena;
  std::vector<int64_t> rec_off;        // first base of every record in the arrays above (a multiple of 64)
  std::vector<int64_t> rec_len;        // bases of every record
  std::vector<int64_t> exc_pos;        // ascending, relative to the file's arrays
  std::vector<uint8_t> exc_val;
  int64_t total = 0;                   // bases incl. padding (a multiple of 64)
  size_t file_bytes = 0;
};

namespace fasta_detail {

inline std::vector<char> &io_buffer() { static thread_local std::vector<char> b; return b; }

// the whole file into the calling thread's buffer, 64 readable zero bytes behind it (the 32-byte loads of the packer)
inline size_t slurp(const char *path, std::vector<char> &buf) {
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) throw Error(FA_ERR_IO, std::string(path) + ": " + strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0) { const int e = errno; ::close(fd); throw Error(FA_ERR_IO, std::string(path) + ": " + strerror(e)); }
  if (S_ISDIR(st.st_mode)) { ::close(fd); throw Error(FA_ERR_IO, std::string(path) + ": is a directory"); }
  // a regular file is read in one sweep of the size fstat reports; a FIFO, a process substitution or a procfs path reports
  // size 0 and is read until end of file instead (it used to come back as a silently empty genome)
  const bool regular = S_ISREG(st.st_mode);
  size_t size = regular ? (size_t)st.st_size : (size_t)(1u << 20), got = 0;
  // the buffer of a pool thread is reused from file to file; one that has grown beyond 256 MB (a multi-gigabase assembly) is
  // given back before the next file, so that two dozen workers do not each pin the largest file they ever read
  if (buf.capacity() > ((size_t)256 << 20) && size + 64 < buf.capacity() / 2) { std::vector<char>().swap(buf); }
  if (buf.size() < size + 64) buf.resize(std::max(size + 64, buf.size() + buf.size() / 2));
  for (;;) {
    if (got == size) {
      if (regular) break;
      size *= 2;                                                    // (not a regular file: keep reading until read() returns 0)
      buf.resize(size + 64);
    }
    const ssize_t r = ::read(fd, buf.data() + got, size - got);
    if (r < 0) { if (errno == EINTR) continue; const int e = errno; ::close(fd); throw Error(FA_ERR_IO, std::string(path) + ": read: " + strerror(e)); }
    if (r == 0) break;                                              // (shorter than fstat said: take what there is)
    got += (size_t)r;
  }
  ::close(fd);
  memset(buf.data() + got, 0, 64);
  return got;
}

// 2-bit codes appended to a word array, any number of bases at a time
struct BitSink {
  uint32_t *dst;
  uint64_t acc = 0;
  int fill = 0;                                                     // bits in acc, < 64
  explicit BitSink(uint32_t *d) : dst(d) {}
  inline void put(uint64_t code, int nbits) {                       // nbits in [0, 64], `code` is zero above them
    acc |= code << fill;
    if (fill + nbits >= 64) {
      memcpy(dst, &acc, 8); dst += 2;
      acc = fill ? code >> (64 - fill) : 0;
      fill += nbits - 64;
    } else fill += nbits;
  }
  // the last partial words, then zeros up to `end`
  inline void finish(uint32_t *end) {
    if (fill > 0) { memcpy(dst, &acc, 8); dst += 2; acc = 0; fill = 0; }
    while (dst < end) *dst++ = 0u;
  }
};

// 32 bytes -> their 2-bit codes, the bytes that are plain nucleotides, the bytes that are newlines (pack32_avx2 with both masks out)
__attribute__((target("avx2"))) inline void classify32(const uint8_t *src, uint64_t &code, uint32_t &plain, uint32_t &newline) {
  const __m256i v = _mm256_loadu_si256((const __m256i *)src);
  const __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), _mm256_set1_epi8(3));
  const __m256i lut = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i back = _mm256_shuffle_epi8(lut, c);
  const __m256i upper = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));
  plain = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(upper, back));
  newline = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('\n')));
  const __m256i pairs = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0401));
  const __m256i quads = _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00100001));
  const __m256i sel = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i by = _mm256_shuffle_epi8(quads, sel);
  code = (uint64_t)(uint32_t)_mm256_extract_epi32(by, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(by, 4) << 32);
}

// m <= 32 bases the general way: table look-up, exceptions recorded at `at` + j
inline uint64_t codes_slow(const uint8_t *p, int m, int64_t at, std::vector<int64_t> &epos, std::vector<uint8_t> &eval) {
  uint64_t code = 0;
  for (int j = 0; j < m; j++) {
    const uint8_t ch = p[j];
    uint8_t c = kCodeOf[ch];
    if (c > 3) { epos.push_back(at + j); eval.push_back(host_upper(ch)); c = 0; }
    code |= (uint64_t)c << (2 * j);
  }
  return code;
}

// the sequence lines [b0, b1) of one record into words from `dst` on; returns the bases; `base` = file-relative offset of its first base
__attribute__((target("avx2"))) inline int64_t pack_body_avx2(const uint8_t *data, size_t b0, size_t b1, uint32_t *dst, int64_t base,
                                                              std::vector<int64_t> &epos, std::vector<uint8_t> &eval) {
  BitSink sink(dst);
  int64_t n = 0;
  size_t p = b0;
  while (p < b1) {
    uint64_t code; uint32_t plain, nl;
    classify32(data + p, code, plain, nl);                          // (64 readable bytes behind the file: slurp)
    const size_t left = b1 - p;
    if (left < 32) nl |= 1u << left;                                // the record ends here: treat its end as a line end
    const int m = nl ? __builtin_ctz(nl) : 32;                      // bases before the next newline
    const uint32_t need = m == 32 ? 0xFFFFFFFFu : ((1u << m) - 1u);
    if ((plain & need) != need) code = codes_slow(data + p, m, base + n, epos, eval);
    else if (m < 32) code &= (1ULL << (2 * m)) - 1ULL;
    sink.put(code, 2 * m);
    n += m;
    p += (size_t)m + ((m < 32 && (size_t)m < left) ? 1u : 0u);      // past the newline, if that is what ended the run
  }
  sink.finish(dst + (n + 63) / 64 * 4);
  return n;
}
inline int64_t pack_body_scalar(const uint8_t *data, size_t b0, size_t b1, uint32_t *dst, int64_t base,
                                std::vector<int64_t> &epos, std::vector<uint8_t> &eval) {
  BitSink sink(dst);
  int64_t n = 0;
  size_t p = b0;
  while (p < b1) {
    const uint8_t *e = (const uint8_t *)memchr(data + p, '\n', b1 - p);
    const size_t end = e ? (size_t)(e - data) : b1;
    for (size_t q = p; q < end; q += 32) {
      const int m = (int)std::min<size_t>(32, end - q);
      sink.put(codes_slow(data + q, m, base + n, epos, eval), 2 * m);
      n += m;
    }
    p = e ? end + 1 : end;
  }
  sink.finish(dst + (n + 63) / 64 * 4);
  return n;
}

}  // namespace fasta_detail

// (FA_TRACE: where a file's task spends its time, summed over the tasks -- nanoseconds of reading, record search, packing)
struct FastaTaskClock { std::atomic<uint64_t> read_ns{0}, scan_ns{0}, pack_ns{0}, files{0}; };
inline FastaTaskClock &fasta_task_clock() { static FastaTaskClock c; return c; }

// The bytes of a file for the one-sweep reader, with 64 readable zero bytes behind them: read() into the thread's buffer, or --
// FA_FASTA_IO=mmap -- the file mapped over the front of an anonymous reservation one page longer than the file (the page
// behind the last file page is the reservation's own zero page, so the 32-byte loads of the packer may run past the end).
struct FileBytes {
  const char *data = nullptr;
  size_t size = 0;
  void *map = nullptr;
  size_t map_len = 0;
  FileBytes() = default;
  FileBytes(const FileBytes &) = delete;
  FileBytes &operator=(const FileBytes &) = delete;
  ~FileBytes() { if (map) munmap(map, map_len); }
  void open(const char *path) {
    static const bool use_mmap = [] { const char *e = getenv("FA_FASTA_IO"); return e && std::string(e) == "mmap"; }();
    if (!use_mmap) {
      std::vector<char> &buf = fasta_detail::io_buffer();
      size = fasta_detail::slurp(path, buf);
      data = buf.data();
      return;
    }
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) throw Error(FA_ERR_IO, std::string(path) + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { const int e = errno; ::close(fd); throw Error(FA_ERR_IO, std::string(path) + ": " + strerror(e)); }
    if (S_ISDIR(st.st_mode)) { ::close(fd); throw Error(FA_ERR_IO, std::string(path) + ": is a directory"); }
    size = (size_t)st.st_size;
    const size_t page = 4096;
    map_len = (size + page - 1) / page * page + page;
    void *r = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (r == MAP_FAILED) { ::close(fd); throw Error(FA_ERR_NOMEM, std::string(path) + ": mmap (reservation) failed"); }
    map = r;
    if (size > 0) {
      void *p = mmap(r, size, PROT_READ, MAP_PRIVATE | MAP_FIXED | MAP_POPULATE, fd, 0);
      if (p == MAP_FAILED) { const int e = errno; ::close(fd); throw Error(FA_ERR_IO, std::string(path) + ": mmap: " + strerror(e)); }
    }
    ::close(fd);
    data = (const char *)r;
  }
};

inline void read_fasta_packed(const char *path, bool protein, PackedFasta &out, const std::shared_ptr<HostArena> &arena = nullptr) {
  using namespace fasta_detail;
  static const bool timed = getenv("FA_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  FileBytes fb;
  fb.open(path);
  const size_t size = fb.size;
  const auto t1 = std::chrono::steady_clock::now();
  FastaFile f;
  f.attach(fb.data, size);
  std::vector<FastaFile::Span> spans;
  FastaFile::Span sp;
  size_t body_bytes = 0;
  while (f.next_span(sp)) { spans.push_back(sp); body_bytes += sp.body_end - sp.body; }
  const auto t2 = std::chrono::steady_clock::now();
  out = PackedFasta();
  out.protein = protein;
  out.file_bytes = size;
  out.rec_off.reserve(spans.size()); out.rec_len.reserve(spans.size());
  // (an upper bound that needs no counting sweep: a body holds at most as many bases as bytes, and every record is padded
  //  to 64 bases)
  const size_t cap_bases = body_bytes + 64 * spans.size() + 64;
  const uint8_t *data = (const uint8_t *)fb.data;
  int64_t at = 0;
  if (protein) {
    out.bytes = arena ? (uint8_t *)arena->take(cap_bases) : nullptr;
    if (out.bytes) out.arena = arena; else { out.own_bytes.reset(new uint8_t[cap_bases]); out.bytes = out.own_bytes.get(); }
    for (const auto &s : spans) {
      uint8_t *dst = out.bytes + at;
      int64_t n = 0;
      size_t p = s.body;
      while (p < s.body_end) {
        const uint8_t *e = (const uint8_t *)memchr(data + p, '\n', s.body_end - p);
        const size_t end = e ? (size_t)(e - data) : s.body_end;
        for (size_t k = p; k < end; k++) dst[n++] = host_upper(data[k]);
        p = e ? end + 1 : end;
      }
      const int64_t padded = (n + 63) / 64 * 64;
      for (int64_t i = n; i < padded; i++) dst[i] = 0;
      out.rec_off.push_back(at); out.rec_len.push_back(n);
      at += padded;
    }
  } else {
    out.words = arena ? (uint32_t *)arena->take((cap_bases / 16 + 4) * 4) : nullptr;
    if (out.words) out.arena = arena; else { out.own_words.reset(new uint32_t[cap_bases / 16 + 4]); out.words = out.own_words.get(); }
    const bool avx2 = host_has_avx2();
    for (const auto &s : spans) {
      uint32_t *dst = out.words + at / 16;
      const int64_t n = avx2 ? pack_body_avx2(data, s.body, s.body_end, dst, at, out.exc_pos, out.exc_val)
                             : pack_body_scalar(data, s.body, s.body_end, dst, at, out.exc_pos, out.exc_val);
      out.rec_off.push_back(at); out.rec_len.push_back(n);
      at += (n + 63) / 64 * 64;
    }
  }
  out.total = at;
  f.close();
  if (timed) {
    const auto t3 = std::chrono::steady_clock::now();
    auto ns = [](auto a, auto b) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
    FastaTaskClock &c = fasta_task_clock();
    c.read_ns += ns(t0, t1); c.scan_ns += ns(t1, t2); c.pack_ns += ns(t2, t3); c.files++;
  }
}

// (FA_FASTA_THREADS=<n>: pool workers that take files; default: the pool's usual two dozen -- on tmpfs 128 readers spent 70 ms
//  per 5 MB file inside read(), 25 readers 3-4 ms)
inline int fasta_pool_helpers() { static const int v = [] { const char *e = getenv("FA_FASTA_THREADS"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 24; }(); return v; }

// Many files at once: one task per file on the pool (a file never waits for the one before it).
inline void read_fasta_packed_many(const char *const *paths, size_t n, bool protein, std::vector<PackedFasta> &out) {
  out.clear();
  out.resize(n);
  // one arena for the packed words of all files, sized from the file sizes (a body holds at most as many bases as bytes; 3 % and
  // a page per file for the padding of records -- a file of very many tiny records that outgrows its share allocates for itself)
  std::shared_ptr<HostArena> arena;
  if (n > 1) {
    size_t need = 0;
    for (size_t i = 0; i < n; i++) {
      struct stat st;
      const size_t sz = stat(paths[i], &st) == 0 && st.st_size > 0 ? (size_t)st.st_size : 0;
      need += (protein ? sz : sz / 4) + sz / 32 + 4096;
    }
    arena = std::make_shared<HostArena>(need);
  }
  HostPool::get().parallel_for(n, [&](size_t i) { read_fasta_packed(paths[i], protein, out[i], arena); }, fasta_pool_helpers());
}


// Records of packed files placed into a sequence store: the counterpart of HostStore::pack_many / append_many for input
// that is packed already.  `use_len` <= the record's length: a query batch holds the whole fragments of a contig only, and a
// prefix of a packed record is a prefix of its words (the tail of the last word is cleared, padding is written as zeros,
// exceptions behind the prefix are dropped).
struct PackedRef { const PackedFasta *file; int64_t rec; int64_t use_len; };

inline int64_t packed_padded_bases(const PackedRef *refs, int64_t n) {
  int64_t add = 0;
  for (int64_t q = 0; q < n; q++) add += (refs[q].use_len + 63) / 64 * 64;
  return add;
}

// into caller memory (dst32: padded bases / 16 words, dst8: padded bases bytes); offsets, lengths and exceptions are appended to `hs`
inline void place_packed(HostStore &hs, const PackedRef *refs, int64_t n, uint32_t *dst32, uint8_t *dst8) {
  std::vector<size_t> at((size_t)n + 1, 0);                          // destination offset of every record, in bases
  hs.seq_off.reserve(hs.seq_off.size() + (size_t)n); hs.seq_len.reserve(hs.seq_len.size() + (size_t)n);
  const int64_t store0 = hs.total;
  for (int64_t q = 0; q < n; q++) {
    hs.seq_off.push_back(store0 + (int64_t)at[q]);
    hs.seq_len.push_back(refs[q].use_len);
    at[q + 1] = at[q] + (size_t)((refs[q].use_len + 63) / 64 * 64);
  }
  hs.total += (int64_t)at[n];
  HostPool::get().parallel_for((size_t)n, [&](size_t q) {
    const PackedRef &r = refs[q];
    const int64_t len = r.use_len, padded = (len + 63) / 64 * 64, src0 = r.file->rec_off[r.rec];
    if (hs.protein) {
      uint8_t *d = dst8 + at[q];
      memcpy(d, r.file->bytes + src0, (size_t)len);
      memset(d + len, 0, (size_t)(padded - len));
    } else {
      uint32_t *d = dst32 + at[q] / 16;
      const uint32_t *src = r.file->words + src0 / 16;
      const int64_t whole = len / 16, rest = len % 16;
      memcpy(d, src, (size_t)whole * 4);
      int64_t w = whole;
      if (rest) d[w++] = src[whole] & ((1u << (2 * rest)) - 1u);
      for (; w < padded / 16; w++) d[w] = 0u;
    }
  });
  for (int64_t q = 0; q < n; q++) {
    const PackedFasta &f = *refs[q].file;
    if (f.exc_pos.empty()) continue;
    const int64_t lo = f.rec_off[refs[q].rec], hi = lo + refs[q].use_len, to = store0 + (int64_t)at[q] - lo;
    auto a = std::lower_bound(f.exc_pos.begin(), f.exc_pos.end(), lo);
    auto b = std::lower_bound(a, f.exc_pos.end(), hi);
    for (auto it = a; it != b; ++it) { hs.exc_pos.push_back(*it + to); hs.exc_val.push_back(f.exc_val[(size_t)(it - f.exc_pos.begin())]); }
  }
}

// appended to the store's own arrays (all or nothing, like HostStore::append_many)
inline void append_packed(HostStore &hs, const PackedRef *refs, int64_t n) {
  const size_t add = (size_t)packed_padded_bases(refs, n);
  auto grow = [](auto &v, size_t need) { if (need > v.capacity()) v.reserve(std::max(need, v.capacity() * 2)); };
  const size_t o_packed = hs.packed.size(), o_bytes = hs.bytes.size(), o_seq = hs.seq_off.size(), o_exc = hs.exc_pos.size();
  const int64_t o_total = hs.total;
  try {
    if (hs.protein) {
      grow(hs.bytes, o_bytes + add);
      hs.bytes.resize(o_bytes + add);
      place_packed(hs, refs, n, nullptr, hs.bytes.data() + o_bytes);
    } else {
      grow(hs.packed, o_packed + add / 16);
      hs.packed.resize(o_packed + add / 16);
      place_packed(hs, refs, n, hs.packed.data() + o_packed, nullptr);
    }
  } catch (...) {
    hs.packed.resize(o_packed); hs.bytes.resize(o_bytes); hs.seq_off.resize(o_seq); hs.seq_len.resize(o_seq);
    hs.exc_pos.resize(o_exc); hs.exc_val.resize(o_exc); hs.total = o_total;
    throw;
  }
}

}  // namespace fa
