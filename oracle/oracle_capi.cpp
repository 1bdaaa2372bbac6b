// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY (see fastani_oracle.hpp).
// Flat C entry points over the CPU restatement so tests/, smoke() and
// bench.py's cpu_baseline leg can drive it through ctypes.
#include "fastani_oracle.hpp"

#include <chrono>

using namespace fo;

struct OracleHandle {
  Sketch sk;
  bool indexed = false;
  std::vector<MappingResult> last_mappings;
  std::vector<CGIResult> last_rows;
  std::vector<Hit> last_hits;
};

extern "C" {

uint32_t fo_hash(const uint8_t *seq, int len) { return get_hash(seq, len); }

int fo_recommended_window(double p_value, int k, int alphabet, float identity, int frag, uint64_t ref_size) {
  return recommended_window_size(p_value, k, alphabet, identity, frag, ref_size);
}
int fo_min_hits_relaxed(int s, int k, float pid) { return estimate_minimum_hits_relaxed(s, k, pid); }
int fo_min_hits(int s, int k, float pid) { return estimate_minimum_hits(s, k, pid); }
float fo_j2md(float j, int k) { return j2md(j, k); }
float fo_md2j(float d, int k) { return md2j(d, k); }
float fo_md_lower_bound(float d, int s, int k, float ci) { return md_lower_bound(d, s, k, ci); }
// identity and upper-bound identity of an L2 mapping with `shared` of `s` sketch elements
void fo_identity(int shared, int s, int k, float *identity, float *upper) {
  float md = j2md((float)(1.0 * shared / s), k);
  float lo = md_lower_bound(md, s, k, L2_CONFIDENCE_INTERVAL);
  *identity = 100 * (1 - md);
  *upper = 100 * (1 - lo);
}

void *fo_new(int k, int frag, float min_fraction, double p_value, float pid, uint64_t ref_size, int protein, int window) {
  OracleHandle *h = new OracleHandle();
  Parameters &p = h->sk.param;
  p.kmerSize = k; p.minReadLength = frag; p.minFraction = min_fraction; p.p_value = p_value;
  p.percentageIdentity = pid; p.referenceSize = ref_size;
  if (protein) { p.alphabetSize = 20; p.windowSize = 1; }
  else { p.alphabetSize = 4; p.windowSize = window > 0 ? window : recommended_window_size(p_value, k, 4, pid, frag, ref_size); }
  return h;
}
void fo_free(void *hh) { delete (OracleHandle *)hh; }
int fo_window(void *hh) { return ((OracleHandle *)hh)->sk.param.windowSize; }

int fo_add_contig(void *hh, const void *data, int64_t len, int width) {
  return ((OracleHandle *)hh)->sk.add_contig(data, width, len);
}
void fo_end_genome(void *hh) { ((OracleHandle *)hh)->sk.end_genome(); }
// several genomes at once, sketched by `threads` host threads (test-speed helper: same records as the calls above)
void fo_add_genomes(void *hh, const void **contigs, const int64_t *lens, const int32_t *contig_genome, int64_t n_contigs,
                    int32_t n_genomes, int width, int threads) {
  std::vector<std::vector<Sketch::ContigRef>> genomes((size_t)n_genomes);
  for (int64_t i = 0; i < n_contigs; i++) genomes[(size_t)contig_genome[i]].push_back(Sketch::ContigRef{contigs[i], width, lens[i]});
  ((OracleHandle *)hh)->sk.add_genomes_parallel(genomes, threads);
}
int64_t fo_num_minimizers(void *hh) { return (int64_t)((OracleHandle *)hh)->sk.minimizerIndex.size(); }
void fo_get_minimizers(void *hh, uint32_t *hash, int32_t *seq, int32_t *wpos) {
  const auto &v = ((OracleHandle *)hh)->sk.minimizerIndex;
  for (size_t i = 0; i < v.size(); i++) { hash[i] = v[i].hash; seq[i] = v[i].seqId; wpos[i] = v[i].wpos; }
}
// minimizers of one stand-alone sequence (query-fragment mode: seqId 0, fresh output)
int64_t fo_sketch_sequence(void *hh, const void *data, int64_t len, int width, uint32_t *hash, int32_t *wpos, int64_t cap) {
  const Parameters &p = ((OracleHandle *)hh)->sk.param;
  std::vector<MinimizerInfo> out;
  add_minimizers(out, data, width, len, p.kmerSize, p.windowSize, 0, p.alphabetSize != 4);
  for (size_t i = 0; i < out.size() && (int64_t)i < cap; i++) { hash[i] = out[i].hash; wpos[i] = out[i].wpos; }
  return (int64_t)out.size();
}

void fo_index(void *hh) {
  OracleHandle *h = (OracleHandle *)hh;
  h->sk.index();
  h->sk.compute_freq_hist();
  h->indexed = true;
}
int fo_freq_threshold(void *hh) { return ((OracleHandle *)hh)->sk.freqThreshold; }
int64_t fo_index_size(void *hh) { return (int64_t)((OracleHandle *)hh)->sk.minimizerPosLookupIndex.size(); }
int64_t fo_index_count(void *hh, uint32_t hash) {
  const auto &m = ((OracleHandle *)hh)->sk.minimizerPosLookupIndex;
  auto it = m.find(hash);
  return it == m.end() ? -1 : (int64_t)it->second.size();
}
int64_t fo_num_genomes(void *hh) { return (int64_t)((OracleHandle *)hh)->sk.lengths.size(); }
uint64_t fo_genome_length(void *hh, int64_t g) { return ((OracleHandle *)hh)->sk.lengths[g]; }

// L1 candidates of one fragment (stage test): returns count, fills up to cap
int fo_l1_fragment(void *hh, const void *frag, int width, int *sketch_size, int *min_hits, int32_t *seq, int32_t *start,
                   int32_t *end, int cap) {
  OracleHandle *h = (OracleHandle *)hh;
  Mapper map(h->sk);
  Query Q; Q.len = h->sk.param.minReadLength;
  std::vector<L1Candidate> l1;
  map.do_l1(frag, width, h->sk.param.minReadLength, Q, l1);
  *sketch_size = Q.sketchSize;
  *min_hits = Q.sketchSize ? estimate_minimum_hits_relaxed(Q.sketchSize, h->sk.param.kmerSize, h->sk.param.percentageIdentity) : 0;
  for (size_t i = 0; i < l1.size() && (int)i < cap; i++) { seq[i] = l1[i].seqId; start[i] = l1[i].rangeStartPos; end[i] = l1[i].rangeEndPos; }
  return (int)l1.size();
}

// Full query.  Returns number of hits; details are fetched with the getters below.
int fo_query(void *hh, const void **contigs, const int64_t *lens, int n, int width, int threads, int *n_short,
             uint64_t *total_fragments, uint64_t *total_length, double *seconds) {
  OracleHandle *h = (OracleHandle *)hh;
  std::vector<ContigView> cv;
  for (int i = 0; i < n; i++) cv.push_back(ContigView{contigs[i], width, lens[i]});
  h->last_hits.clear(); h->last_mappings.clear(); h->last_rows.clear();
  auto t0 = std::chrono::steady_clock::now();
  query_draft(h->sk, h->sk.lengths, cv, threads, h->last_hits, n_short, &h->last_mappings, &h->last_rows, total_fragments,
              total_length);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  return (int)h->last_hits.size();
}
void fo_get_hits(void *hh, int32_t *genome, float *identity, int32_t *matches, int32_t *fragments) {
  const auto &v = ((OracleHandle *)hh)->last_hits;
  for (size_t i = 0; i < v.size(); i++) { genome[i] = v[i].refGenomeId; identity[i] = v[i].identity; matches[i] = v[i].matches; fragments[i] = v[i].fragments; }
}
int64_t fo_num_rows(void *hh) { return (int64_t)((OracleHandle *)hh)->last_rows.size(); }
void fo_get_rows(void *hh, int32_t *genome, float *identity, int32_t *count) {
  const auto &v = ((OracleHandle *)hh)->last_rows;
  for (size_t i = 0; i < v.size(); i++) { genome[i] = v[i].refGenomeId; identity[i] = v[i].identity; count[i] = v[i].countSeq; }
}
int64_t fo_num_mappings(void *hh) { return (int64_t)((OracleHandle *)hh)->last_mappings.size(); }
void fo_get_mappings(void *hh, int32_t *qseq, int32_t *rseq, int32_t *rstart, int32_t *sketch, int32_t *shared, float *identity) {
  const auto &v = ((OracleHandle *)hh)->last_mappings;
  for (size_t i = 0; i < v.size(); i++) {
    qseq[i] = v[i].querySeqId; rseq[i] = v[i].refSeqId; rstart[i] = v[i].refStartPos;
    sketch[i] = v[i].sketchSize; shared[i] = v[i].conservedSketches; identity[i] = v[i].nucIdentity;
  }
}

}  // extern "C"
