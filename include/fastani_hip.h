/* include/fastani_hip.h
 *
 * C ABI of libfastani_hip.so: the MI355X-native FastANI fragment-mapping
 * engine that stands in for the C++ symbols pyfastani's Cython module
 * cimports (there is no plugin registry in the reference; the boundary IS
 * that cimport surface).  Every entry point names the reference
 * interface it replaces (paths relative to the pyfastani checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; opaque handles own all device memory;
 *   - every function returns 0 on success, non-zero on failure; the message
 *     is available from fa_last_error() (thread-local); nothing throws
 *     across the boundary (reference: `except +` / `except 1 nogil`,
 *     include/fastani/map/compute_map.pxd:31-36, src/pyfastani/_fastani.pyx:164,893);
 *   - input buffers are borrowed for the duration of the call
 *     (_fastani.pyx:1095 memoryview over caller memory);
 *   - contigs are passed as (pointer, length, char_width) with char_width
 *     1, 2 or 4 = the PyUnicode kind (_fastani.pyx:633-645,1073-1092);
 *   - there is NO CPU fallback: without a HIP device the compute entry points
 *     fail with FA_ERR_NO_DEVICE.
 */
#ifndef FASTANI_HIP_H
#define FASTANI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_OK 0
#define FA_ERR_INVALID 1     /* bad argument (maps to ValueError)              */
#define FA_ERR_NO_DEVICE 2   /* no HIP device / HIP runtime error (RuntimeError) */
#define FA_ERR_NOMEM 3       /* host or device allocation failed (MemoryError)  */
#define FA_ERR_UNSUPPORTED 4 /* parameter regime outside the HIP path           */
#define FA_ERR_INTERNAL 5
#define FA_ERR_IO 6          /* file could not be opened / mapped (OSError)      */
#define FA_ERR_BUFFER 7      /* FASTA header longer than the reference's line buffer (BufferError) */

typedef struct fa_sketch fa_sketch;   /* skch::Sketch under construction + pyfastani bookkeeping */
typedef struct fa_mapper fa_mapper;   /* indexed reference, resident in HBM (skch::Sketch after index() + skch::Map) */
typedef struct fa_genomes fa_genomes; /* a batch of query genomes packed 2-bit and resident in HBM */

/* skch::Parameters, include/fastani/map/map_parameters.pxd:9-24 (fields the path reads) */
typedef struct fa_params {
  int32_t kmer_size;            /* kmerSize            */
  int32_t window_size;          /* windowSize          */
  int32_t fragment_length;      /* minReadLength       */
  int32_t alphabet_size;        /* alphabetSize: 4 nucleotide, 20 protein */
  float min_fraction;           /* minFraction         */
  float percentage_identity;    /* percentageIdentity  */
  double p_value;               /* p_value             */
  uint64_t reference_size;      /* referenceSize       */
} fa_params;

/* cgi::CGI_Results, include/fastani/cgi/cgid_types.pxd:19-27, plus the query index of a batch */
typedef struct fa_cgi_row {
  int32_t query_id;               /* position of the query genome in the batch (qryGenomeId) */
  int32_t ref_genome_id;          /* refGenomeId          */
  int32_t count_seq;              /* countSeq  -> Hit.matches   */
  int32_t total_query_fragments;  /* totalQueryFragments -> Hit.fragments */
  float identity;                 /* identity  -> Hit.identity  */
} fa_cgi_row;

/* one L2 mapping, include/fastani/map/base_types.pxd:52-63 (fields consumed downstream) */
typedef struct fa_mapping {
  int32_t query_seq_id;     /* querySeqId: fragment number inside its query genome */
  int32_t ref_seq_id;       /* refSeqId   */
  int32_t ref_start_pos;    /* refStartPos (= meanOptimalPos) */
  int32_t sketch_size;      /* sketchSize */
  int32_t conserved;        /* conservedSketches */
  int32_t query_id;
} fa_mapping;

/* ---- library ---------------------------------------------------------- */
const char *fa_last_error(void);
int fa_version(void);
/* Device memory the library's handles give back is kept in a per-device pool and reused (on this runtime memory that went
 * through hipFree is scrubbed before it is handed out again, which made the second index build of a process twelve times
 * slower than the first).  fa_device_trim returns everything the pool holds to the runtime (*held_bytes, if not NULL: what
 * it held); FA_POOL_MAX_GB (default 96, 0 = no pool) bounds it. */
int fa_device_trim(uint64_t *held_bytes);
int fa_device_count(int *count);
int fa_set_device(int device);            /* one process per GPU: call once per rank */

/* ---- parameter statistics (host) -------------------------------------- */
/* skch::Stat::recommendedWindowSize, include/fastani/map/map_stats.pxd:22-29, called _fastani.pyx:553-560.
 * *window = -1 when no sketch size reaches the p-value cut-off (upstream reads an uninitialised value there). */
int fa_recommended_window_size(double p_value, int k, int alphabet_size, float identity, int fragment_length,
                               uint64_t reference_size, int *window);
/* skch::Stat::estimateMinimumHitsRelaxed, map_stats.pxd:11, called _fastani.pyx:951 */
int fa_estimate_minimum_hits_relaxed(int sketch_size, int k, float identity, int *hits);
/* nucIdentity / nucIdentityUpperBound of skch::Map::doL2Mapping for (shared, sketch_size) */
int fa_mapping_identity(int shared, int sketch_size, int k, float *identity, float *upper_bound);
/* skch::CommonFunc::getHash, include/fastani/map/common_func.pxd:12 (host twin of the device function) */
uint32_t fa_hash(const void *kmer, int length);

/* ---- Sketch: reference side ------------------------------------------- */
/* `new Sketch_t(param)`, _fastani.pyx:476 */
int fa_sketch_new(const fa_params *params, fa_sketch **out);
void fa_sketch_free(fa_sketch *s);
/* one iteration of the contig loop of Sketch._add_draft, _fastani.pyx:629-683 (addMinimizers on device, lazily).
 * *added = 0 when the contig is shorter than the window or k-mer size (the UserWarning case, :670-677). */
int fa_sketch_add_contig(fa_sketch *s, const void *data, int64_t length, int char_width, int *added);
/* tail of Sketch._add_draft, _fastani.pyx:686-690: closes the genome, records its fragment-rounded length */
int fa_sketch_end_genome(fa_sketch *s);
/* the exception path of Sketch._add_draft: the reference sums the genome length in a local (`total`, _fastani.pyx:618,680)
 * that is lost when a contig raises half-way, while the contigs already added stay in the sketch; call this from the
 * binding's exception handler so that the abandoned genome's length is not carried into the next one. */
int fa_sketch_abort_genome(fa_sketch *s);
/* Sketch.clear, _fastani.pyx:746-767 */
int fa_sketch_clear(fa_sketch *s);
/* len(Sketch.minimizers) / Minimizers.__getitem__, _fastani.pyx:1222-1235 (device -> host read-back) */
int fa_sketch_num_minimizers(fa_sketch *s, int64_t *n);
int fa_sketch_get_minimizers(fa_sketch *s, uint32_t *hash, int32_t *seq_id, int32_t *wpos);
/* Sketch.__getstate__/__setstate__, _fastani.pyx:572-591 */
int fa_sketch_num_genomes(fa_sketch *s, int64_t *n);
int fa_sketch_get_state(fa_sketch *s, uint64_t *lengths, int32_t *sequences_by_file, int64_t *counter);
int fa_sketch_set_state(fa_sketch *s, int64_t n_genomes, const uint64_t *lengths, const int32_t *sequences_by_file,
                        int64_t counter, int64_t n_minimizers, const uint32_t *hash, const int32_t *seq_id,
                        const int32_t *wpos);
/* Device-pointer variants of fa_sketch_get_minimizers / fa_sketch_set_state (same record layout as the pickled state,
 * _fastani.pyx:572-591): the three arrays live in HBM buffers owned by the caller (e.g. torch tensors), so the
 * multi-GPU index build can all-gather minimizer shards over RCCL without a host round trip.
 * `cap` is the capacity of the destination arrays in records. */
int fa_sketch_get_minimizers_device(fa_sketch *s, int64_t cap, uint32_t *d_hash, int32_t *d_seq_id, int32_t *d_wpos);
int fa_sketch_set_state_device(fa_sketch *s, int64_t n_genomes, const uint64_t *lengths, const int32_t *sequences_by_file,
                               int64_t counter, int64_t n_minimizers, const uint32_t *d_hash, const int32_t *d_seq_id,
                               const int32_t *d_wpos);
/* Sketch.index, _fastani.pyx:769-806: Sketch_t::index() + computeFreqHist(); ownership of the data moves to the
 * mapper and the sketch is left cleared but usable. */
int fa_sketch_index(fa_sketch *s, fa_mapper **out);

/* ---- Mapper: query side ------------------------------------------------ */
void fa_mapper_free(fa_mapper *m);
/* Sketch_t::getFreqThreshold, include/fastani/map/win_sketch.pxd:40 */
int fa_mapper_freq_threshold(fa_mapper *m, int *threshold);
/* len(Mapper.lookup_index) = minimizerPosLookupIndex.size(), _fastani.pyx:1454-1456 */
int fa_mapper_lookup_size(fa_mapper *m, int64_t *n);
/* The HIP device the mapper's index lives on (fa_set_device at the time Sketch.index() ran; -1: the calling thread's
 * current device).  No reference counterpart (the reference has no device); used by the multi-GPU layer to allocate the
 * tensors it hands to fa_mapper_lookup_export_device / fa_mapper_set_global_frequency on the right GPU. */
int fa_mapper_device(fa_mapper *m, int *device);
/* Reference-sharded index ("when the index does not fit"): every rank indexes its own share of
 * the reference genomes, and the frequency threshold of Sketch_t::computeFreqHist / the `size < threshold` filter of
 * _fastani.pyx:946 must then be taken over the position lists of ALL shards.  fa_mapper_lookup_export_device copies
 * the distinct hashes of this shard (ascending) and their list lengths into caller-owned HBM buffers (e.g. torch
 * tensors, `cap` >= fa_mapper_lookup_size) for the exchange; fa_mapper_set_global_frequency installs the threshold
 * computed over all shards and the hashes (device array) whose summed list length reaches it: the lookup ignores
 * them from then on although their local lists are short.  Call it before the first query on this mapper. */
int fa_mapper_lookup_export_device(fa_mapper *m, int64_t cap, uint32_t *d_keys, int32_t *d_counts);
int fa_mapper_set_global_frequency(fa_mapper *m, int threshold, int64_t n_drop, const uint32_t *d_drop_keys);
/* MinimizerIndex.__iter__/__getitem__, _fastani.pyx:1458-1475 */
int fa_mapper_lookup_keys(fa_mapper *m, uint32_t *keys);
int fa_mapper_lookup_count(fa_mapper *m, uint32_t hash, int64_t *count); /* -1 when absent */
int fa_mapper_lookup_get(fa_mapper *m, uint32_t hash, int32_t *seq_id, int32_t *wpos, int64_t cap);
int fa_mapper_num_minimizers(fa_mapper *m, int64_t *n);
int fa_mapper_get_minimizers(fa_mapper *m, uint32_t *hash, int32_t *seq_id, int32_t *wpos);
int fa_mapper_num_genomes(fa_mapper *m, int64_t *n);
int fa_mapper_get_state(fa_mapper *m, uint64_t *lengths, int32_t *sequences_by_file);

/* Mapper._query_draft up to and including computeCGI, _fastani.pyx:1006-1118, for ONE query genome given as
 * host buffers.  rows receive one cgi::CGI_Results per reference genome with at least one mapping, in
 * refGenomeId order; the minimum_fraction filter and the sort (_fastani.pyx:1121-1136) stay with the caller,
 * which owns the names.  *n_short = contigs skipped with the short-sequence warning (:1061-1070).
 * Re-entrant like the reference's query (:1158-1161): every call borrows one of the mapper's workspaces (a HIP stream
 * and every intermediate buffer), so calls from several host threads on ONE mapper overlap on the device; the stage
 * getters and fa_mapper_last_timings below report the most recently finished call. */
int fa_mapper_query(fa_mapper *m, const void *const *contigs, const int64_t *lengths, int n_contigs, int char_width,
                    fa_cgi_row *rows, int64_t cap, int64_t *n_rows, int *n_short, uint64_t *total_fragments,
                    uint64_t *total_length);

/* ---- host ingest: FASTA files ------------------------------------------ */
/* Record reader with the semantics of pyfastani._fasta.Parser (src/pyfastani/_fasta.pyx:41-103): records exist only
 * if the first line starts with '>'; id = header line without '>' and newline; sequence lines joined, ASCII letters
 * upper-cased (copy_upper); a header that does not end in '\n' within 2047 bytes fails with FA_ERR_BUFFER.  The
 * pointers returned by fa_fasta_next stay valid until the next call on the same handle. */
typedef struct fa_fasta fa_fasta;
int fa_fasta_open(const char *path, fa_fasta **out);                          /* Parser.__cinit__, _fasta.pyx:49-58 */
int fa_fasta_next(fa_fasta *f, int *has_record, const char **id, int64_t *id_length, const unsigned char **seq,
                  int64_t *seq_length);                                         /* Parser.__next__, _fasta.pyx:66-103 */
void fa_fasta_close(fa_fasta *f);
/* Parser + Sketch._add_draft (_fastani.pyx:610-690) in one native call: every record of the file is a contig of ONE
 * reference genome; records are split and upper-cased by host threads and packed without passing through Python. */
int fa_sketch_add_fasta(fa_sketch *s, const char *path, int64_t *n_records, int64_t *n_short);
/* n_genomes reference genomes from host buffers in ONE call: contig c belongs to genome contig_genome[c] (non-decreasing,
 * < n_genomes; a genome may have no contig); equivalent to fa_sketch_add_contig for every contig and fa_sketch_end_genome
 * for every genome (_fastani.pyx:610-690 per genome), with one run of the packer over all contigs.  n_short: [n_genomes]
 * contigs skipped with the short-sequence warning, or NULL. */
int fa_sketch_add_genomes(fa_sketch *s, const void *const *contigs, const int64_t *lengths, const int32_t *contig_genome,
                          int64_t n_contigs, int32_t n_genomes, int char_width, int32_t *n_short);
/* The same for n_paths reference genomes, one per file, in the order given: the files are read and 2-bit packed
 * concurrently (one host task per file, straight from the file's bytes to packed words), then added as n_paths
 * consecutive fa_sketch_add_fasta calls would have added them.  n_records / n_short: [n_paths] or NULL.  The host side
 * of the reference's benchmark loop `for path: sketch.add_draft(name, [r.seq for r in Parser(path)])`
 * (benches/mapping/bench.py:41-47). */
int fa_sketch_add_fasta_many(fa_sketch *s, const char *const *paths, int32_t n_paths, int64_t *n_records, int64_t *n_short);
/* One query genome per FASTA file, packed and uploaded as a resident batch (fa_genomes_upload semantics); the files
 * are read concurrently, and the upload runs on a stream of the batch's own. */
int fa_genomes_upload_fasta(fa_mapper *m, const char *const *paths, int32_t n_paths, fa_genomes **out);
/* FASTA files read and 2-bit packed ONCE (one genome per file, the files concurrently), to be used as references and as
 * queries: an all-vs-all -- the reference benchmark's shape, benches/mapping/bench.py:41-53: the genomes that are sketched are
 * the genomes that are mapped -- reads every file one time.  protein != 0: residue bytes are kept instead. */
typedef struct fa_packed fa_packed;
int fa_packed_read(const char *const *paths, int32_t n_paths, int protein, fa_packed **out);
/* more files behind the ones it holds.  Thread-safe against the calls that read the set (info, add_packed, reload_packed): it
 * grows the set under an exclusive lock they hold shared; fa_packed_free must not race with any of them. */
int fa_packed_append(fa_packed *p, const char *const *paths, int32_t n_paths);
void fa_packed_free(fa_packed *p);
/* per file: its size in bytes, its records, its bases (arrays of *n_files entries, any may be NULL) */
int fa_packed_info(fa_packed *p, int32_t *n_files, uint64_t *file_bytes, int64_t *records, int64_t *bases);
/* files [first, first + count) as that many reference genomes (fa_sketch_add_fasta_many without reading) */
int fa_sketch_add_packed(fa_sketch *s, fa_packed *p, int32_t first, int32_t count, int64_t *n_records, int64_t *n_short);
/* files [first, first + count) as the query genomes of a recycled batch (fa_genomes_reload_fasta without reading) */
int fa_genomes_reload_packed(fa_mapper *m, fa_genomes *g, fa_packed *p, int32_t first, int32_t count);
/* Refills a batch from other files, recycling its device buffers, its pinned staging image and its stream: the
 * double-buffered form for a stream of query chunks (while one batch is mapped by fa_mapper_query_genomes on one host
 * thread, another thread refills the other batch).  A failed refill leaves an empty batch. */
int fa_genomes_reload_fasta(fa_mapper *m, fa_genomes *g, const char *const *paths, int32_t n_paths);

/* ---- resident batches (many-to-many; inputs stay in HBM) -------------- */
/* Pack + upload a batch of query genomes.  contig_genome[i] is the genome (0..n_genomes-1, non-decreasing)
 * contig i belongs to.  Short contigs are skipped exactly as _fastani.pyx:1061-1070 does. */
int fa_genomes_upload(fa_mapper *m, const void *const *contigs, const int64_t *lengths, const int32_t *contig_genome,
                      int64_t n_contigs, int32_t n_genomes, int char_width, fa_genomes **out);
void fa_genomes_free(fa_genomes *g);
int fa_genomes_info(fa_genomes *g, int32_t *n_genomes, uint64_t *total_fragments, uint64_t *total_length,
                    int32_t *n_short); /* arrays of n_genomes entries, may be NULL */
/* Map genomes [first, first+count) of a resident batch against the resident index: the hot path
 * (K1 sketch -> lookup -> L1 -> L2 -> CGI), everything on device.  rows as fa_mapper_query, query_id = index in
 * the batch.  If rows_device is non-zero, `rows` is a DEVICE pointer (e.g. a torch tensor feeding an RCCL
 * all-gather) with room for `cap` rows. */
int fa_mapper_query_genomes(fa_mapper *m, fa_genomes *g, int32_t first, int32_t count, fa_cgi_row *rows, int64_t cap,
                            int64_t *n_rows, int rows_device);

/* stage-level introspection used by the parity tests */
int fa_mapper_debug_mappings(fa_mapper *m, fa_mapping *out, int64_t cap, int64_t *n); /* L2 results of the last query call */
int fa_mapper_debug_l1(fa_mapper *m, int32_t *frag, int32_t *seq_id, int32_t *range_start, int32_t *range_end,
                       int64_t cap, int64_t *n);
int fa_mapper_debug_query_sketch(fa_mapper *m, int64_t fragment, uint32_t *hashes, int32_t cap, int32_t *sketch_size);
/* winnowed minimizers of one stand-alone sequence in query-fragment mode (seqId 0, fresh output vector) */
int fa_debug_sketch_sequence(const fa_params *params, const void *data, int64_t length, int char_width,
                             uint32_t *hash, int32_t *wpos, int64_t cap, int64_t *n);

/* development probe: workgroups of 128 threads with `lds_bytes` of dynamic LDS the chip holds at once */
int fa_debug_probe_occupancy(int lds_bytes, int *peak_alive);
/* raw bytes of the last call's event arena (two-kernel L2 form: the slide events; FA_FUSED_DEBUG=8: per-workgroup time
 * stamps of k_l2_fused) -- development aid */
int fa_mapper_debug_items(fa_mapper *m, void *out, int64_t bytes);
/* the query-independent slide geometry the index build derives per reference record: rec_prev,
 * rec_fwd, rec_bwd (4 bytes each) and the flag byte, `cap` records at most; *n = records of the index.  Checked against the
 * definitions by the parity tests (there is no counterpart in the reference: slidingMap.hpp keeps a std::map instead) */
int fa_mapper_debug_links(fa_mapper *m, int32_t *prev, int32_t *fwd, int32_t *bwd, uint8_t *flags, int64_t cap, int64_t *n);
/* slide events per L2 locus of the last call (two-kernel form), in locus order -- development aid */
int fa_mapper_debug_locus_events(fa_mapper *m, uint32_t *events, int64_t cap, int64_t *n);
/* last-call statistics: [0] sketch ms (K1 + fragment sort/unique), [1] lookup + L1 ms, [2] L2 ms, [3] CGI ms,
 * [4] total ms -- device time between stamps of the chip-wide 100 MHz counter that the first kernel of every stage
 * leaves in the pass's status block (the kernels of a pass run back to back on the library's stream) -- then counters
 * of the call: [5] reference records inside L2 locus ranges, [6] L2 loci, [7] L2 slide events, [8] loci redone with
 * the wide L2 state, [9] passes repeated because a speculated buffer size was too small; after fa_mapper_query also
 * the host-side wall-clock split of that call: [10] packing ms, [11] fragment / tile tables ms, [12] uploads ms,
 * [13] device pass + rows ms; [14], [15] development (fused L2 form); [16] the L2 stage once more, bracketed by HIP
 * events on the library's stream, when fa_mapper_set_stage_events is on (0 otherwise); [17] parts of the call whose sketch
 * stage ran as ONE launch (k_query_fused), [18] parts that ran K1 and the fragment sketch as two kernels, [19] parts whose
 * k_l2_events workgroups ran in the offset-major order; [23] positions per tile of the last fa_bench_sketch_kernel.  n <= 24. */
int fa_mapper_last_timings(fa_mapper *m, float *ms, int n);
/* on != 0: also bracket the L2 stage of every pass with two HIP events (slot [16] above).  Off by default: an event
 * record costs the stream about as much as a small kernel. */
int fa_mapper_set_stage_events(fa_mapper *m, int on);
/* the HIP stream the library launches on (so callers can bracket it with their own events) */
int fa_mapper_stream(fa_mapper *m, void **stream);
/* run only the minimizer-extraction kernel (K1) over a resident batch `repeat` times and report the mean
 * kernel time; used by bench.py for the roofline line.  The batch is sketched the way REFERENCE genomes are
 * (_fastani.pyx:651-659: whole contigs, windows across fragment boundaries): its fragments are joined into the contigs
 * they were cut from and tiled as fa_sketch tiles them. */
int fa_bench_sketch_kernel(fa_mapper *m, fa_genomes *g, int repeat, float *ms_per_launch, uint64_t *bases,
                           uint64_t *minimizers);

#ifdef __cplusplus
}
#endif
#endif /* FASTANI_HIP_H */
