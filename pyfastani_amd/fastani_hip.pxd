# cython: language_level=3
# Cython declarations of the C ABI of libfastani_hip.so (include/fastani_hip.h).
#
# This is the file a pyfastani maintainer adds next to include/fastani/**/*.pxd: every entry names the reference
# interface it stands in for (paths relative to the pyfastani checkout; the boundary of the reference is the Cython
# cimport of FastANI's C++ symbols, src/pyfastani/_fastani.pyx:51-84).
from libc.stdint cimport int32_t, int64_t, uint32_t, uint64_t

cdef extern from "fastani_hip.h" nogil:
    ctypedef struct fa_sketch
    ctypedef struct fa_mapper
    ctypedef struct fa_genomes
    ctypedef struct fa_fasta

    ctypedef struct fa_params:          # skch::Parameters, include/fastani/map/map_parameters.pxd:9-24
        int32_t kmer_size
        int32_t window_size
        int32_t fragment_length
        int32_t alphabet_size
        float min_fraction
        float percentage_identity
        double p_value
        uint64_t reference_size

    ctypedef struct fa_cgi_row:         # cgi::CGI_Results, include/fastani/cgi/cgid_types.pxd:19-27
        int32_t query_id
        int32_t ref_genome_id
        int32_t count_seq
        int32_t total_query_fragments
        float identity

    ctypedef struct fa_mapping:         # MappingResult, include/fastani/map/base_types.pxd:52-63
        int32_t query_seq_id
        int32_t ref_seq_id
        int32_t ref_start_pos
        int32_t sketch_size
        int32_t conserved
        int32_t query_id

    const char* fa_last_error()
    int fa_version()
    int fa_device_trim(uint64_t* held_bytes)
    int fa_device_count(int* count)
    int fa_set_device(int device)

    # skch::Stat, include/fastani/map/map_stats.pxd:6-29
    int fa_recommended_window_size(double p_value, int k, int alphabet_size, float identity, int fragment_length,
                                   uint64_t reference_size, int* window)                      # _fastani.pyx:553-560
    int fa_estimate_minimum_hits_relaxed(int sketch_size, int k, float identity, int* hits)   # :951
    uint32_t fa_hash(const void* kmer, int length)                                            # getHash, common_func.pxd:12

    # Sketch_t under construction + pyfastani's counters
    int fa_sketch_new(const fa_params* params, fa_sketch** out)                               # `new Sketch_t(param)`, :476
    void fa_sketch_free(fa_sketch* s)                                                         # :570
    int fa_sketch_add_contig(fa_sketch* s, const void* data, int64_t length, int char_width, int* added)   # :629-683
    int fa_sketch_end_genome(fa_sketch* s)                                                    # :686-690
    int fa_sketch_abort_genome(fa_sketch* s)                                                  # exception path of :610-690 (`total` is a local there)
    int fa_sketch_clear(fa_sketch* s)                                                         # :755-765
    int fa_sketch_num_minimizers(fa_sketch* s, int64_t* n)                                    # :1223
    int fa_sketch_get_minimizers(fa_sketch* s, uint32_t* hash, int32_t* seq_id, int32_t* wpos)   # :1225-1254
    int fa_sketch_num_genomes(fa_sketch* s, int64_t* n)
    int fa_sketch_get_state(fa_sketch* s, uint64_t* lengths, int32_t* sequences_by_file, int64_t* counter)   # :572-582
    int fa_sketch_set_state(fa_sketch* s, int64_t n_genomes, const uint64_t* lengths, const int32_t* sequences_by_file,
                            int64_t counter, int64_t n_minimizers, const uint32_t* hash, const int32_t* seq_id,
                            const int32_t* wpos)                                              # :584-591
    int fa_sketch_get_minimizers_device(fa_sketch* s, int64_t cap, uint32_t* d_hash, int32_t* d_seq_id, int32_t* d_wpos)
    int fa_sketch_set_state_device(fa_sketch* s, int64_t n_genomes, const uint64_t* lengths, const int32_t* sequences_by_file,
                                   int64_t counter, int64_t n_minimizers, const uint32_t* d_hash, const int32_t* d_seq_id,
                                   const int32_t* d_wpos)
    int fa_sketch_add_fasta(fa_sketch* s, const char* path, int64_t* n_records, int64_t* n_short)   # _fasta.pyx:41-103 + :610-690
    int fa_sketch_add_genomes(fa_sketch* s, const void* const* contigs, const int64_t* lengths, const int32_t* contig_genome, int64_t n_contigs, int32_t n_genomes, int char_width, int32_t* n_short)
    int fa_sketch_add_fasta_many(fa_sketch* s, const char* const* paths, int32_t n_paths, int64_t* n_records, int64_t* n_short)
    int fa_sketch_index(fa_sketch* s, fa_mapper** out)                                        # :790-791 (+ ownership move :793-806)

    # Sketch_t after index() + skch::Map
    void fa_mapper_free(fa_mapper* m)                                                         # :839-840
    int fa_mapper_freq_threshold(fa_mapper* m, int* threshold)                                # getFreqThreshold, :600
    int fa_mapper_lookup_size(fa_mapper* m, int64_t* n)                                       # :1456
    int fa_mapper_device(fa_mapper* m, int* device)
    int fa_mapper_lookup_export_device(fa_mapper* m, int64_t cap, uint32_t* d_keys, int32_t* d_counts)
    int fa_mapper_set_global_frequency(fa_mapper* m, int threshold, int64_t n_drop, const uint32_t* d_drop_keys)
    int fa_mapper_lookup_keys(fa_mapper* m, uint32_t* keys)                                   # :1458-1466
    int fa_mapper_lookup_count(fa_mapper* m, uint32_t hash, int64_t* count)                   # :1468-1475
    int fa_mapper_lookup_get(fa_mapper* m, uint32_t hash, int32_t* seq_id, int32_t* wpos, int64_t cap)
    int fa_mapper_num_minimizers(fa_mapper* m, int64_t* n)
    int fa_mapper_get_minimizers(fa_mapper* m, uint32_t* hash, int32_t* seq_id, int32_t* wpos)
    int fa_mapper_num_genomes(fa_mapper* m, int64_t* n)
    int fa_mapper_get_state(fa_mapper* m, uint64_t* lengths, int32_t* sequences_by_file)      # :842-851
    int fa_mapper_query(fa_mapper* m, const void* const* contigs, const int64_t* lengths, int n_contigs, int char_width,
                        fa_cgi_row* rows, int64_t cap, int64_t* n_rows, int* n_short, uint64_t* total_fragments,
                        uint64_t* total_length)                                               # body of _query_draft, :1052-1118

    # resident batches (many-to-many extension)
    int fa_genomes_upload(fa_mapper* m, const void* const* contigs, const int64_t* lengths, const int32_t* contig_genome,
                          int64_t n_contigs, int32_t n_genomes, int char_width, fa_genomes** out)
    int fa_genomes_upload_fasta(fa_mapper* m, const char* const* paths, int32_t n_paths, fa_genomes** out)
    ctypedef struct fa_packed:
        pass
    int fa_packed_read(const char* const* paths, int32_t n_paths, int protein, fa_packed** out)
    int fa_packed_append(fa_packed* p, const char* const* paths, int32_t n_paths)
    void fa_packed_free(fa_packed* p)
    int fa_packed_info(fa_packed* p, int32_t* n_files, uint64_t* file_bytes, int64_t* records, int64_t* bases)
    int fa_sketch_add_packed(fa_sketch* s, fa_packed* p, int32_t first, int32_t count, int64_t* n_records, int64_t* n_short)
    int fa_genomes_reload_packed(fa_mapper* m, fa_genomes* g, fa_packed* p, int32_t first, int32_t count)
    int fa_genomes_reload_fasta(fa_mapper* m, fa_genomes* g, const char* const* paths, int32_t n_paths)
    void fa_genomes_free(fa_genomes* g)
    int fa_genomes_info(fa_genomes* g, int32_t* n_genomes, uint64_t* total_fragments, uint64_t* total_length, int32_t* n_short)
    int fa_mapper_query_genomes(fa_mapper* m, fa_genomes* g, int32_t first, int32_t count, fa_cgi_row* rows, int64_t cap,
                                int64_t* n_rows, int rows_device)
