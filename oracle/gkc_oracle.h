/* gkc_oracle.h — CPU restatement (ORACLE) of GATB-Core's DSK k-mer-counting hot path.
 *
 * THIS IS TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link/load it, and only as the checker. The product (libgkc_hip.so) never calls into it.
 *
 * Every function cites the reference file:line (relative to /root/reference/gatb-core/) it restates.
 * Pinning: tests/test_oracle_golden.py checks this file against every known-answer vector the
 * reference's own unit tests hold for the path (TestDSK, TestKmer, TestMath, TestDebloom, TestContainer).
 * The reference itself is NOT buildable here under the build rules (needs cmake-generated
 * system/api/config.hpp, cmake-generated template specialisations and the cmake-built vendored HDF5),
 * so there is no oracle/_ref; see DESIGN.md "Oracle".
 */
#ifndef GKC_ORACLE_H
#define GKC_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 gko_u128;

/* ---- A1: nucleotide encoding (tools/misc/api/Data.hpp:185, Data.cpp:3) ---- */
int gko_nt_code(unsigned char c);   /* (c>>1)&3 : A=0 C=1 T=2 G=3 */
int gko_nt_valid(unsigned char c);  /* 1 iff c in ACGTacgt */

/* ---- tools/math arithmetic (LargeInt1.pri:137-211, NativeInt64.hpp:210-221, LargeInt2.pri:168-251) ---- */
uint64_t gko_revcomp64(uint64_t x, unsigned k);
uint64_t gko_hash64(uint64_t key, uint64_t seed);
uint64_t gko_oahash64(uint64_t key);
uint64_t gko_simplehash16_li1(uint64_t key, int shift);  /* LargeInt<1> variant: 3 table terms   */
uint64_t gko_simplehash16_ni64(uint64_t key, int shift); /* NativeInt64 / LargeInt<2>: 2 terms   */
void     gko_revcomp128(uint64_t lo, uint64_t hi, unsigned k, uint64_t* out_lo, uint64_t* out_hi);
uint64_t gko_hash1_128(uint64_t lo, uint64_t hi, uint64_t seed);

/* ---- A2: canonical k-mers of one sequence (Model.hpp:637-657, 726-765, 858-884, :294) ----
 * out_* arrays have len-k+1 entries (nothing written if len<k). Returns number of k-mer positions. */
int64_t gko_kmers(const char* seq, uint64_t len, unsigned k,
                  uint64_t* fwd_lo, uint64_t* fwd_hi, uint64_t* can_lo, uint64_t* can_hi, uint8_t* valid);

/* ---- A3: minimizer model (Model.hpp:1012-1064 LUT, :1220-1251 is_allowed, :1107-1139 next,
 *          :1254-1287 computeNewMinimizerOriginal, :957-973 comparator) ----
 * lut has 4^m u32 entries. freq_order NULL => lexicographic/KMC2 mode. */
void    gko_mmer_lut(unsigned m, int has_freq, uint32_t* lut);
int64_t gko_minimizers(const char* seq, uint64_t len, unsigned k, unsigned m,
                       const uint32_t* freq_order, uint32_t* out_minimizer, uint8_t* out_valid);

/* ---- A4: super-k-mer split (Sequence2SuperKmer.hpp:81-159). maxs<=0 => reference default
 *          min((8*sizeof(Type)-8)/2,255) = 28 (k<=31) / 60 (k<=63).
 * Writes up to cap entries of (minimizer, first k-mer index, nbK). Returns number of super-k-mers. */
int64_t gko_superkmers(const char* seq, uint64_t len, unsigned k, unsigned m, const uint32_t* freq_order,
                       int maxs, uint32_t* sk_minimizer, uint32_t* sk_start, uint32_t* sk_nbk, uint64_t cap,
                       uint64_t* n_valid, uint64_t* n_invalid);

/* ---- A6: super-k-mer wire format (Model.hpp:1386-1471 save; Storage.cpp:567-580 insertSuperkmer) ----
 * Encodes [u8 nbK][payload]; returns bytes written. */
size_t gko_superkmer_encode(const char* seq_at_first_kmer, unsigned k, unsigned nbk, uint8_t* out);
/* ---- B1: decode one record and regenerate canonical k-mers (PartitionsCommand.cpp:944-1128) ----
 * Returns bytes consumed; writes nbK canonical k-mers. */
size_t gko_superkmer_decode(const uint8_t* rec, unsigned k, uint64_t* can_lo, uint64_t* can_hi, unsigned* nbk);

/* ---- Repartitor (PartiInfo.cpp:48-218; RepartitionAlgorithm.cpp:311-384) ---- */
/* minimizer frequency order from m-mer counts (computeFrequencies): returns freq_order[4^m] */
void gko_freq_order_from_counts(unsigned m, const uint32_t* mmer_counts, uint32_t* freq_order);
/* m-mer counting functor used by computeFrequencies (RepartitionAlgorithm.cpp:60-120 MmersFrequency):
 * counts canonical m-mers of every VALID m-mer position of the sequence. */
void gko_count_mmers(const char* seq, uint64_t len, unsigned m, uint32_t* mmer_counts);
/* repartition tables; nb_kmers_per_minim / nb_kxmers_per_minim are 4^m u64 sample statistics */
void gko_repart_compute_distrib(unsigned m, uint32_t nb_part, const uint64_t* nb_kxmers_per_minim, uint16_t* table);
void gko_repart_just_group_lexi(unsigned m, uint32_t nb_part, const uint64_t* nb_kmers_per_minim, uint16_t* table);
void gko_repart_just_group(unsigned m, uint32_t nb_part, const uint64_t* nb_kmers_per_minim,
                           const uint32_t* mmer_counts, uint16_t* table);

/* ---- whole path: SortingCountAlgorithm::execute (SortingCountAlgorithm.cpp:636-781) ----
 * reads: flat ASCII buffer + offsets[n_reads+1]. repart: u16[4^m]. freq_order NULL => lexi.
 * Result handle holds, per output dataset (part + pass*nb_partitions), the ascending (value,abundance) list of
 * SOLID k-mers (abundance_min <= count <= abundance_max), plus statistics. */
typedef struct gko_dsk gko_dsk;
gko_dsk* gko_dsk_run(const char* bases, const uint64_t* offsets, uint64_t n_reads,
                     unsigned k, unsigned m, uint32_t nb_partitions, uint32_t nb_passes,
                     const uint16_t* repart, const uint32_t* freq_order,
                     int32_t abundance_min, int32_t abundance_max, uint32_t histo_max, int maxs);
/* the same run on n_threads pthreads, parallelised like the reference (fillPartitions over the reads, fillSolidKmers over the partitions):
 * identical results; this is what bench.py times as cpu_baseline ("port") on all host cores */
gko_dsk* gko_dsk_run_mt(const char* bases, const uint64_t* offsets, uint64_t n_reads,
                        unsigned k, unsigned m, uint32_t nb_partitions, uint32_t nb_passes,
                        const uint16_t* repart, const uint32_t* freq_order,
                        int32_t abundance_min, int32_t abundance_max, uint32_t histo_max, int maxs, uint32_t n_threads);
/* restricted to the partitions with keep[p] != 0 (NULL: all): exact count of a few sampled partitions of a full-size input (see gkc_oracle.c) */
gko_dsk* gko_dsk_run_parts(const char* bases, const uint64_t* offsets, uint64_t n_reads,
                           unsigned k, unsigned m, uint32_t nb_partitions, uint32_t nb_passes,
                           const uint16_t* repart, const uint32_t* freq_order,
                           int32_t amin, int32_t amax, uint32_t histo_max, int maxs, uint32_t n_threads, const uint8_t* keep);
void     gko_dsk_free(gko_dsk*);
uint64_t gko_dsk_part_size(const gko_dsk*, uint32_t dataset);
/* copies dataset as (lo, hi, abundance) arrays */
void     gko_dsk_part_copy(const gko_dsk*, uint32_t dataset, uint64_t* lo, uint64_t* hi, int32_t* abundance);
/* copies dataset in the reference's in-memory Count layout: 16 B {u64 value; i32 abundance; pad} for k<=31,
 * 32 B {u128 value; i32 abundance; pad} for k<=63 (tools/misc/api/Abundance.hpp:68-129) */
void     gko_dsk_part_copy_records(const gko_dsk*, uint32_t dataset, void* out);
/* stats[0]=kmers_nb_valid [1]=kmers_nb_invalid [2]=kmers_nb_distinct [3]=kmers_nb_solid
 * [4]=nb_superkmers [5]=nb_sequences [6]=superkmer bytes in reference wire format [7]=sequences shorter than k */
void     gko_dsk_stats(const gko_dsk*, uint64_t stats[8]);
/* histogram of abundances over ALL distinct k-mers, histo_max+1 bins (Histogram.hpp:92) */
void     gko_dsk_histogram(const gko_dsk*, uint64_t* histo);
/* per-partition super-k-mer statistics of pass 0..: n_kmers / n_superkmers per dataset */
void     gko_dsk_part_stats(const gko_dsk*, uint32_t dataset, uint64_t* n_kmers, uint64_t* n_superkmers);

/* ---- C1-C4: Bloom filters (tools/collections/impl/Bloom.hpp:59-98, 170-300, 386-828, 1240-1282) ---- */
enum { GKO_BLOOM_BASIC = 0, GKO_BLOOM_CACHE = 1, GKO_BLOOM_NEIGHBOR = 2 };
typedef struct gko_bloom gko_bloom;
gko_bloom* gko_bloom_create(int kind, uint64_t tai_bits, unsigned nb_hash, unsigned k);
void       gko_bloom_free(gko_bloom*);
uint64_t   gko_bloom_nbytes(const gko_bloom*);         /* getSize()    */
uint64_t   gko_bloom_bitsize(const gko_bloom*);        /* getBitSize() */
uint8_t*   gko_bloom_array(gko_bloom*);                /* getArray()   */
void       gko_bloom_seeds(uint64_t user_seed, uint64_t out[10]);   /* HashFunctors::generate_hash_seed */
void       gko_bloom_insert(gko_bloom*, const uint64_t* lo, const uint64_t* hi, uint64_t n);
void       gko_bloom_contains(const gko_bloom*, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* out);
/* neighbor kind only: contains8 bitset (bits 0-3 right ext A,C,T,G ; 4-7 left ext) */
void       gko_bloom_contains8(const gko_bloom*, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif

/* ---- bank: FASTA / FASTQ text -> sequences (BankFasta::Iterator::get_next_seq_from_file, bank/impl/BankFasta.cpp:488-571;
 *      buffered_gets :425-483). Restated on an in-memory text: same character-level state machine (header = first token + rest of
 *      line, sequence lines appended verbatim up to '\n' with ONE trailing '\r' dropped when the accumulated read is longer than 1,
 *      a line starting with '>' '@' ends the sequence, '+' starts the quality which is consumed BY LENGTH, then everything up to the
 *      next '>' / '@' character is skipped). Output: flat data + n_seq+1 offsets; returns n_seq, or -1 if a capacity is too small. */
int64_t gko_fastx_parse(const char* text, uint64_t n, char* out_data, uint64_t cap_data, uint64_t* out_offsets, uint64_t cap_seq);

/* ---- Histogram::compute_threshold (tools/misc/impl/Histogram.cpp:61-190): histo[0..length] -> out = {cutoff, nbsolids, first_peak} */
void gko_histogram_cutoff(const uint64_t* histo, uint64_t length, int min_auto_threshold, uint64_t out[3]);

/* ---- MPHF: BooPHF (thirdparty/BooPHF/BooPHF.h:714-1200) as GATB instantiates it (tools/collections/impl/BooPHF.hpp:236-300:
 *      jenkins64 hasher seeded by std::mt19937_64(37), gamma = 3, 25 levels) + the abundance map of MPHFAlgorithm::populate
 *      (kmer/impl/MPHFAlgorithm.cpp:222-275) with MapMPHF's discretization table (tools/collections/impl/MapMPHF.hpp:96-145).
 *      Keys: n items of `stride` bytes whose first 8 (wide = 0) or 16 (wide = 1) bytes are the k-mer, little-endian. */
typedef struct gko_mphf gko_mphf;
gko_mphf* gko_mphf_build(const void* keys, uint64_t n, uint32_t stride, int wide);
void      gko_mphf_free(gko_mphf*);
uint64_t  gko_mphf_lookup(const gko_mphf*, uint64_t lo, uint64_t hi);        /* ULLONG_MAX: not in the set (final level miss) */
uint64_t  gko_mphf_save(const gko_mphf*, uint8_t* out, uint64_t cap);         /* mphf::save byte stream; returns its size (out may be NULL) */
int       gko_abundance_index(int abundance);                                 /* MPHFAlgorithm.cpp:253-266 */
