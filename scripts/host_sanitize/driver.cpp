// Host-side pieces of libfastani_hip under AddressSanitizer + UBSan, and (separately) ThreadSanitizer, on the CPU:
//   the 2-bit packer (fa_host.h: AVX2 path, scalar exception path, wide characters, protein bytes) on the persistent thread
//   pool, the memory-mapped FASTA reader (fa_fasta.h), the statistics tables (fa_stats.h), the workspace lease and the
//   pinned-word spin (fa_lease.h).  Inputs: the edge cases of tests/test_gpu_parity.py::test_minimizer_streams and
//   tests/test_fasta.py, plus four concurrent clients.  No HIP: these headers are what fa_engine.hip includes for the same jobs.
// Built and run by scripts/host_sanitize.sh; exits non-zero on any mismatch (the sanitizers abort on their own findings).
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../pyfastani_amd/csrc/fa_fasta.h"
#include "../../pyfastani_amd/csrc/fa_host.h"
#include "../../pyfastani_amd/csrc/fa_lease.h"
#include "../../pyfastani_amd/csrc/fa_stats.h"

using namespace fa;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

// ---- reference packer: the definition of the store, byte by byte ----
struct Plain { std::vector<uint32_t> packed; std::vector<uint8_t> bytes; std::vector<int64_t> epos; std::vector<uint8_t> eval; std::vector<int64_t> off; int64_t total = 0; };
static uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
static void plain_append(Plain &p, bool protein, const std::vector<uint32_t> &seq) {   // seq as code points (any width)
  const int64_t len = (int64_t)seq.size(), padded = (len + 63) / 64 * 64;
  p.off.push_back(p.total);
  if (protein) {
    for (int64_t i = 0; i < padded; i++) p.bytes.push_back(i < len ? up((uint8_t)seq[i]) : 0);
  } else {
    const size_t w0 = p.packed.size();
    p.packed.resize(w0 + padded / 16, 0u);
    for (int64_t i = 0; i < len; i++) {
      const uint8_t c = up((uint8_t)seq[i]);
      int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
      if (code < 0) { p.epos.push_back(p.total + i); p.eval.push_back(c); code = 0; }
      p.packed[w0 + i / 16] |= (uint32_t)code << (2 * (i % 16));
    }
  }
  p.total += padded;
}

template <class A, class B> static bool same_vec(const A &a, const B &b) { return a.size() == b.size() && std::equal(a.begin(), a.end(), b.begin()); }
template <class T> static std::vector<T> widen(const std::vector<uint32_t> &s) { return std::vector<T>(s.begin(), s.end()); }

static std::vector<std::vector<uint32_t>> edge_sequences(std::mt19937_64 &rng) {
  auto str = [](const std::string &s) { return std::vector<uint32_t>(s.begin(), s.end()); };
  std::vector<std::vector<uint32_t>> v;
  v.push_back({});
  v.push_back(str("A"));
  v.push_back(str("ACGTACGTACGTACG"));                       // 15: one partial word
  v.push_back(str("ACGTACGTACGTACGT"));                      // 16
  v.push_back(str("acgtacgtacgtacgtN"));                     // lower case + an exception in the tail word
  v.push_back(str(std::string(31, 'T') + "N" + std::string(33, 'g')));
  v.push_back(str(std::string(200, 'N')));                   // exceptions only
  v.push_back(str("ACGTRYKMSWBDHVNacgtrykmswbdhvn*-." + std::string(70, 'C')));   // IUPAC, both cases, punctuation
  const char alpha[] = "ACGTacgtNnRYKM";
  for (int len : {17, 63, 64, 65, 127, 128, 1000, 4097, 70001, 300017}) {
    std::vector<uint32_t> s((size_t)len);
    const int exc_every = len > 5000 ? 3001 : 37;            // long runs of plain bases (the AVX2 path) with rare exceptions
    for (int i = 0; i < len; i++) s[(size_t)i] = (i % exc_every == exc_every - 1) ? (uint32_t)alpha[8 + rng() % 6] : (uint32_t)alpha[rng() % 8];
    v.push_back(std::move(s));
  }
  return v;
}

static void test_packer(bool protein, int width, int clients) {
  std::mt19937_64 rng(1234 + width + (protein ? 100 : 0));
  const auto seqs = edge_sequences(rng);
  auto one_client = [&](int id) {
    HostStore hs; hs.protein = protein;
    Plain want;
    // contig by contig, then several at once (append_many), as Sketch.add_draft and upload_genomes do
    std::vector<std::vector<uint8_t>> s8; std::vector<std::vector<uint16_t>> s16; std::vector<std::vector<uint32_t>> s32;
    std::vector<const void *> ptrs; std::vector<int64_t> lens;
    for (auto &s : seqs) {
      if (width == 1) { s8.push_back(widen<uint8_t>(s)); ptrs.push_back(s8.back().data()); }
      else if (width == 2) { s16.push_back(widen<uint16_t>(s)); ptrs.push_back(s16.back().data()); }
      else { s32.push_back(s); ptrs.push_back(s32.back().data()); }
      lens.push_back((int64_t)s.size());
    }
    for (size_t i = 0; i < seqs.size() / 2; i++) { hs.append(ptrs[i], width, lens[i]); plain_append(want, protein, seqs[i]); }
    const size_t rest = seqs.size() - seqs.size() / 2;
    hs.append_many(ptrs.data() + seqs.size() / 2, lens.data() + seqs.size() / 2, (int64_t)rest, width);
    for (size_t i = seqs.size() / 2; i < seqs.size(); i++) plain_append(want, protein, seqs[i]);
    CHECK(hs.total == want.total, "client %d: store length %lld vs %lld", id, (long long)hs.total, (long long)want.total);
    CHECK(hs.seq_off == want.off, "client %d: sequence offsets differ", id);
    if (protein) CHECK(same_vec(hs.bytes, want.bytes), "client %d: protein bytes differ", id);
    else {
      CHECK(same_vec(hs.packed, want.packed), "client %d: packed words differ (width %d)", id, width);
      CHECK(hs.exc_pos == want.epos && hs.exc_val == want.eval, "client %d: exception lists differ (%zu vs %zu)", id, hs.exc_pos.size(), want.epos.size());
    }
    // pack_many into caller memory that is exactly as large as promised (ASan guards its ends)
    HostStore h2; h2.protein = protein;
    const int64_t add = HostStore::padded_bases(lens.data(), (int64_t)lens.size());
    std::vector<uint32_t> d32(protein ? 0 : (size_t)add / 16);
    std::vector<uint8_t> d8(protein ? (size_t)add : 0);
    h2.pack_many(ptrs.data(), lens.data(), (int64_t)lens.size(), width, protein ? nullptr : d32.data(), protein ? d8.data() : nullptr);
    if (protein) CHECK(d8 == want.bytes, "client %d: pack_many bytes differ", id); else CHECK(d32 == want.packed, "client %d: pack_many words differ", id);
  };
  std::vector<std::thread> th;
  for (int c = 0; c < clients; c++) th.emplace_back(one_client, c);
  for (auto &t : th) t.join();
}

static std::string write_tmp(const std::string &name, const std::string &content) {
  char dir[] = "/tmp/fa_sanitize_XXXXXX";
  static std::string base = mkdtemp(dir);
  const std::string path = base + "/" + name;
  FILE *f = fopen(path.c_str(), "wb");
  fwrite(content.data(), 1, content.size(), f);
  fclose(f);
  return path;
}

static void test_fasta(int clients) {
  struct Case { std::string name, text; std::vector<std::pair<std::string, std::string>> want; bool buffer_error; };
  std::string big;                                              // one record of > 256 KiB: cut into pieces at line starts
  std::string big_seq;
  for (int i = 0; i < 9000; i++) { std::string line(60, "acgtn"[i % 5]); big += line + "\n"; for (char c : line) big_seq += (char)up((uint8_t)c); }
  std::vector<Case> cases = {
    {"empty.fa", "", {}, false},
    {"noheader.fa", "ACGT\n>x\nAC\n", {}, false},                // first line is not a header: no records at all
    {"one.fa", ">id one\nACgt\nNNac\n", {{"id one", "ACGTNNAC"}}, false},
    {"noeol.fa", ">a\nAC\n>b\nGT", {{"a", "AC"}, {"b", "GT"}}, false},
    {"emptyrec.fa", ">a\n>b\n\nAC\n\n>c\n", {{"a", ""}, {"b", "AC"}, {"c", ""}}, false},
    {"gt_inside.fa", ">a\nAC>GT\nTT\n", {{"a", "AC>GTTT"}}, false}, // '>' that is not at a line start belongs to the body
    {"big.fa", ">big\n" + big + ">tail\nAC\n", {{"big", big_seq}, {"tail", "AC"}}, false},
    {"longid.fa", ">" + std::string(3000, 'x') + "\nAC\n", {}, true},
    {"headeronly_noeol.fa", ">abc", {}, true},                   // the reference insists on the newline
  };
  auto one_client = [&](int id) {
    for (auto &c : cases) {
      const std::string path = write_tmp(std::to_string(id) + "_" + c.name, c.text);
      bool threw = false;
      std::vector<std::pair<std::string, std::string>> got;
      try {
        FastaFile f; f.open(path.c_str());
        while (f.next()) got.emplace_back(f.id, std::string(f.seq.begin(), f.seq.end()));
      } catch (const Error &e) { threw = e.code == FA_ERR_BUFFER; }
      CHECK(threw == c.buffer_error, "%s: buffer error %d, expected %d", c.name.c_str(), (int)threw, (int)c.buffer_error);
      if (!c.buffer_error) CHECK(got == c.want, "%s: records differ (%zu vs %zu)", c.name.c_str(), got.size(), c.want.size());
      threw = false;
      std::vector<FastaSeq> seqs;
      try { read_fasta_records(path.c_str(), seqs); } catch (const Error &e) { threw = e.code == FA_ERR_BUFFER; }
      CHECK(threw == c.buffer_error, "%s (bulk): buffer error %d, expected %d", c.name.c_str(), (int)threw, (int)c.buffer_error);
      if (!c.buffer_error) {
        CHECK(seqs.size() == c.want.size(), "%s (bulk): %zu records vs %zu", c.name.c_str(), seqs.size(), c.want.size());
        for (size_t r = 0; r < seqs.size() && r < c.want.size(); r++)
          CHECK(std::string((const char *)seqs[r].data.get(), seqs[r].size) == c.want[r].second, "%s (bulk): record %zu differs", c.name.c_str(), r);
      }
      unlink(path.c_str());
    }
    bool io = false;
    try { FastaFile f; f.open("/nonexistent/fa.fa"); } catch (const Error &e) { io = e.code == FA_ERR_IO; }
    CHECK(io, "a missing file should fail with FA_ERR_IO");
    io = false;
    try { FastaFile f; f.open("/tmp"); } catch (const Error &e) { io = e.code == FA_ERR_IO; }
    CHECK(io, "a directory should fail with FA_ERR_IO");
  };
  std::vector<std::thread> th;
  for (int c = 0; c < clients; c++) th.emplace_back(one_client, c);
  for (auto &t : th) t.join();
}

// read_fasta_packed (one sweep from file bytes to 2-bit words) and place_packed against the definition: the records of
// FastaFile, packed byte by byte; whole records and prefixes (a query batch holds whole fragments only); many files at once
static void test_fasta_packed(bool protein) {
  std::mt19937_64 rng(protein ? 77 : 78);
  std::vector<std::string> texts = {
    "", "ACGT\n>x\nAC\n", ">id one\nACgt\nNNac\n", ">a\nAC\n>b\nGT", ">a\n>b\n\nAC\n\n>c\n", ">a\nAC>GT\nTT\n", ">crlf\r\nACGT\r\nAC\r\n",
    ">x\n" + std::string(31, 'A') + "\n" + std::string(32, 'C') + "\n" + std::string(33, 'G') + "\n" + std::string(64, 'T') + "\n" + std::string(65, 'a') + "\nN",
  };
  const char alpha[] = "ACGTACGTACGTACGTacgtNnRYKM*";
  for (int t = 0; t < 40; t++) {                                   // random files: 1-6 records, line widths 1-100, rare exceptions
    std::string text;
    const int nrec = 1 + (int)(rng() % 6);
    for (int r = 0; r < nrec; r++) {
      text += ">rec" + std::to_string(r) + " some description\n";
      const size_t len = (rng() % 5 == 0) ? rng() % 70 : rng() % 20000;
      const size_t width = 1 + rng() % 100;
      const bool dirty = rng() % 3 == 0;
      for (size_t i = 0; i < len; i++) {
        text += dirty && rng() % 50 == 0 ? alpha[rng() % (sizeof(alpha) - 1)] : "ACGT"[rng() % 4];
        if ((i + 1) % width == 0) text += '\n';
      }
      if (rng() % 4) text += '\n';
      if (rng() % 8 == 0) text += "\n\n";
    }
    texts.push_back(text);
  }
  std::vector<std::string> paths;
  for (size_t i = 0; i < texts.size(); i++) paths.push_back(write_tmp("packed_" + std::to_string(i) + ".fa", texts[i]));
  std::vector<const char *> cpaths;
  for (auto &p : paths) cpaths.push_back(p.c_str());
  std::vector<PackedFasta> files;
  read_fasta_packed_many(cpaths.data(), cpaths.size(), protein, files);
  HostStore whole, part;
  whole.protein = part.protein = protein;
  Plain want_whole, want_part;
  std::vector<PackedRef> refs_whole, refs_part;
  for (size_t i = 0; i < texts.size(); i++) {
    FastaFile f; f.open(paths[i].c_str());
    size_t r = 0;
    while (f.next()) {
      CHECK(r < files[i].rec_len.size() && files[i].rec_len[r] == (int64_t)f.seq.size(), "file %zu record %zu: %lld bases vs %zu", i, r,
            r < files[i].rec_len.size() ? (long long)files[i].rec_len[r] : -1LL, f.seq.size());
      if (r >= files[i].rec_len.size()) break;
      std::vector<uint32_t> seq(f.seq.begin(), f.seq.end());
      plain_append(want_whole, protein, seq);
      refs_whole.push_back(PackedRef{&files[i], (int64_t)r, (int64_t)seq.size()});
      const int64_t cut = seq.empty() ? 0 : (int64_t)(rng() % (seq.size() + 1));
      seq.resize((size_t)cut);
      plain_append(want_part, protein, seq);
      refs_part.push_back(PackedRef{&files[i], (int64_t)r, cut});
      r++;
    }
    CHECK(r == files[i].rec_len.size(), "file %zu: %zu records vs %zu", i, files[i].rec_len.size(), r);
  }
  append_packed(whole, refs_whole.data(), (int64_t)refs_whole.size());
  append_packed(part, refs_part.data(), (int64_t)refs_part.size());
  auto same = [&](const HostStore &hs, const Plain &w, const char *what) {
    CHECK(hs.total == w.total, "%s: total %lld vs %lld", what, (long long)hs.total, (long long)w.total);
    CHECK(hs.seq_off == w.off, "%s: sequence offsets differ", what);
    if (protein) CHECK(same_vec(hs.bytes, w.bytes), "%s: bytes differ", what); else CHECK(same_vec(hs.packed, w.packed), "%s: packed words differ", what);
    CHECK(hs.exc_pos == w.epos && hs.exc_val == w.eval, "%s: exceptions differ (%zu vs %zu)", what, hs.exc_pos.size(), w.epos.size());
  };
  same(whole, want_whole, protein ? "packed fasta, protein, whole records" : "packed fasta, whole records");
  same(part, want_part, protein ? "packed fasta, protein, prefixes" : "packed fasta, prefixes");
  bool threw = false;
  const std::string longid = write_tmp("packed_longid.fa", ">" + std::string(3000, 'x') + "\nAC\n");
  try { PackedFasta pf; read_fasta_packed(longid.c_str(), protein, pf); } catch (const Error &e) { threw = e.code == FA_ERR_BUFFER; }
  CHECK(threw, "over-long identifier should fail with FA_ERR_BUFFER in the packed reader");
  threw = false;
  const char *missing[2] = {paths[2].c_str(), "/nonexistent/fa.fa"};
  try { std::vector<PackedFasta> v; read_fasta_packed_many(missing, 2, protein, v); } catch (const Error &e) { threw = e.code == FA_ERR_IO; }
  CHECK(threw, "a missing file among many should fail with FA_ERR_IO");
  unlink(longid.c_str());
  for (auto &p : paths) unlink(p.c_str());
}

static void test_stats() {
  CHECK(stat_recommended_window(1e-3, 16, 4, 80.0f, 3000, 5000000ULL) == 24, "default window is 24 (test_ani.py:60)");
  StatTables t; t.k = 16; t.pid = 80.0f; t.smax = -1;
  CHECK(t.extend(64), "tables grow");
  CHECK(!t.extend(32), "tables never shrink");
  CHECK(t.extend(300), "tables grow again");
  CHECK((int)t.min_hits.size() >= 301 && (int)t.pass_shared.size() >= 301, "table sizes");
  for (int s = 1; s <= 300; s += 13) {
    CHECK(t.min_hits[(size_t)s] == stat_min_hits_relaxed(s, 16, 80.0f), "minHits[%d]", s);
    float id = 0, upper = 0;
    stat_identity(s / 2, s, 16, &id, &upper);
    CHECK(id >= 0.0f && id <= 100.0f && upper >= id, "identity(%d, %d) = %f <= %f", s / 2, s, id, upper);
  }
  for (int k : {3, 5, 14, 16, 21}) for (int s : {1, 2, 17, 263}) { (void)stat_min_hits_relaxed(s, k, 80.0f); (void)stat_min_hits_relaxed(s, k, 99.9f); }
}

struct FakeWs { bool in_use = false; int prepared = 0; std::atomic<int> users{0}; };
struct FakeOwner { static constexpr int NWS = 4; std::mutex mtx; std::condition_variable ws_free; FakeWs ws[NWS]; int last_ws = 0; };

static void test_lease(int threads) {
  FakeOwner m;
  std::atomic<int> thrown{0}, served{0};
  auto client = [&](int id) {
    for (int it = 0; it < 400; it++) {
      try {
        Lease<FakeOwner, FakeWs> l(m, [&](FakeWs &w) { if ((id + it) % 37 == 0) throw Error(FA_ERR_NO_DEVICE, "stream"); w.prepared++; });
        CHECK(l.w->users.fetch_add(1) == 0, "two calls on one workspace");
        if (it % 16 == 0) std::this_thread::yield();
        l.w->users.fetch_sub(1);
        served++;
      } catch (const Error &) { thrown++; }
    }
  };
  std::vector<std::thread> th;
  for (int c = 0; c < threads; c++) th.emplace_back(client, c);
  for (auto &t : th) t.join();
  CHECK(served + thrown == threads * 400 && thrown > 0, "every call either ran or failed in prepare (%d + %d)", served.load(), thrown.load());
  for (auto &w : m.ws) CHECK(!w.in_use, "a workspace was not handed back");
}

static void test_spin() {
  alignas(64) uint32_t word = 0;
  uint32_t payload = 0;
  std::thread pub([&] { std::this_thread::sleep_for(std::chrono::milliseconds(3)); payload = 77; __atomic_store_n(&word, 5u, __ATOMIC_RELEASE); });
  const bool ok = spin_for_seq(&word, 5u, 2000000);
  CHECK(ok && payload == 77, "the released word and what was written before it");
  pub.join();
  CHECK(!spin_for_seq(&word, 6u, 200), "a word that never comes times out");
  CHECK(!spin_for_seq(&word, 6u, 0) && spin_for_seq(&word, 5u, 0), "no spinning: one look");
}

int main() {
  for (int width : {1, 2, 4}) test_packer(false, width, 1);
  test_packer(true, 1, 1);
  test_packer(false, 1, 4);                                      // four concurrent clients on the one thread pool
  test_packer(true, 4, 4);
  test_fasta(1);
  test_fasta(4);
  test_fasta_packed(false);
  test_fasta_packed(true);
  test_stats();
  test_lease(8);
  test_spin();
  // an item that throws inside the pool reaches the caller, and the pool keeps working afterwards
  bool caught = false;
  try { HostPool::get().parallel_for(64, [](size_t i) { if (i == 13) throw Error(FA_ERR_NOMEM, "item"); }); } catch (const Error &e) { caught = e.code == FA_ERR_NOMEM; }
  CHECK(caught, "an exception inside a pool item is re-raised on the caller");
  std::atomic<size_t> sum{0};
  HostPool::get().parallel_for(1000, [&](size_t i) { sum += i; });
  CHECK(sum == 499500, "pool after the exception");
  if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  printf("host pieces: all checks passed\n");
  return 0;
}
