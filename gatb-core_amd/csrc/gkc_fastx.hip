// gkc_fastx.hip — FASTA / FASTQ text -> flat bases + read offsets on gfx950 (the input side of the hot path).
//
// Replaces (reference, under /root/reference/gatb-core/src/gatb/):
//   BankFasta::Iterator::get_next_seq_from_file   bank/impl/BankFasta.cpp:488-571   (kseq-style character state machine)
//   buffered_gets                                  bank/impl/BankFasta.cpp:425-483   (line append + one trailing '\r' dropped)
// and the per-sequence copy into the flat buffer that feeds gkc_push_reads.
//
// The reference reader is a sequential state machine; for WELL-FORMED files its result is a function of the line structure:
//   FASTA : a line whose first character is '>' or '@' is a header; every other line is appended to the current sequence up to its
//           '\n'; one trailing '\r' is dropped when the accumulated sequence is longer than 1 character (BankFasta.cpp:479);
//   FASTQ : records of exactly four lines (header, sequence, '+', quality). The reference consumes the quality BY LENGTH
//           (BankFasta.cpp:556): that equals "one line" iff the quality line is at least as long as the sequence line.
// Everything the line model cannot express exactly (a sequence line starting with '+' in a FASTA file, multi-line FASTQ, a quality
// shorter than its sequence, '>' / '@' inside the lines before the first header) is REFUSED with GKC_ERR_FORMAT — no approximation, no
// host fallback. Parity: tests/test_gpu_fastx.py against the CPU restatement of the character state machine, which is pinned on the
// reference's own bank fixtures (tests/golden/bank, known answers of test/unit/src/bank/TestBank.cpp).
//
// Device algorithm (all passes stream the text with 16-byte loads, 4096-byte tiles, 256 threads):
//   1 newline count per tile -> exclusive scan -> line index at every tile start
//   2 line table: start offset of every line
//   3 one thread per line: kind (skip / header / sequence) + "drop the trailing \r" flag + format checks
//   4 kept-byte count per tile -> exclusive scan; header count per line -> exclusive scan (record ids)
//   5 compaction: kept bytes -> bases (staged through LDS, written in order), offsets[record] at every header line
#include "gkc_common.hpp"
#include "gkc_device.hpp"
#include <vector>

namespace {

constexpr int FX_THREADS = 256, FX_PER = 16, FX_TILE = FX_THREADS * FX_PER;

// ---------------------------------------------------------------------------------------------- generic exclusive scan (u64)
constexpr int FXS_ITEMS = 8, FXS_CHUNK = 1024 * FXS_ITEMS;
__device__ __forceinline__ void fx_wg_scan(uint64_t tv, uint64_t& excl, uint64_t& total, uint64_t* s_w)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint64_t x = tv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up((unsigned long long)x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint64_t p = 0; total = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) p += s_w[w]; total += s_w[w]; }
    excl = p + x - tv;
}
__global__ __launch_bounds__(1024) void k_fx_scan_chunks(uint64_t* __restrict__ a, uint64_t n, uint64_t* __restrict__ ca)
{
    __shared__ uint64_t s_w[16];
    const uint64_t i0 = (uint64_t)blockIdx.x * FXS_CHUNK + (uint64_t)threadIdx.x * FXS_ITEMS;
    uint64_t v[FXS_ITEMS], tv = 0;
#pragma unroll
    for (int j = 0; j < FXS_ITEMS; j++) { v[j] = i0 + j < n ? a[i0 + j] : 0; tv += v[j]; }
    uint64_t r, tot; fx_wg_scan(tv, r, tot, s_w);
#pragma unroll
    for (int j = 0; j < FXS_ITEMS; j++) if (i0 + j < n) { a[i0 + j] = r; r += v[j]; }
    if (threadIdx.x == 0) ca[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_fx_scan_totals(uint64_t* __restrict__ ca, uint32_t n_chunks, uint64_t* __restrict__ total)
{
    __shared__ uint64_t s_w[16];
    const uint32_t i0 = threadIdx.x * FXS_ITEMS;
    uint64_t v[FXS_ITEMS], tv = 0;
#pragma unroll
    for (int j = 0; j < FXS_ITEMS; j++) { v[j] = i0 + j < n_chunks ? ca[i0 + j] : 0; tv += v[j]; }
    uint64_t r, tot; fx_wg_scan(tv, r, tot, s_w);
#pragma unroll
    for (int j = 0; j < FXS_ITEMS; j++) if (i0 + j < n_chunks) { ca[i0 + j] = r; r += v[j]; }
    if (threadIdx.x == 0) *total = tot;
}
__global__ __launch_bounds__(1024) void k_fx_scan_add(uint64_t* __restrict__ a, uint64_t n, const uint64_t* __restrict__ ca)
{
    const uint64_t o = ca[blockIdx.x];
    const uint64_t i0 = (uint64_t)blockIdx.x * FXS_CHUNK + (uint64_t)threadIdx.x * FXS_ITEMS;
#pragma unroll
    for (int j = 0; j < FXS_ITEMS; j++) if (i0 + j < n) a[i0 + j] += o;
}
// in-place exclusive scan of a[0..n); *d_total (device) receives the sum. n <= 2^26.
int fx_scan(gkc_ctx* c, uint64_t* a, uint64_t n, uint64_t* d_total, DevBuf& scratch)
{
    const uint32_t n_chunks = (uint32_t)((n + FXS_CHUNK - 1) / FXS_CHUNK);
    if (n_chunks > (uint32_t)FXS_CHUNK) GKC_FAIL(c, GKC_ERR_ARG, "text chunk too large for one parse call");
    GKC_TRY(c->ensure(scratch, (size_t)std::max<uint32_t>(n_chunks, 1) * 8));
    if (n_chunks) hipLaunchKernelGGL(k_fx_scan_chunks, dim3(n_chunks), dim3(1024), 0, c->stream, a, n, (uint64_t*)scratch.p);
    hipLaunchKernelGGL(k_fx_scan_totals, dim3(1), dim3(1024), 0, c->stream, (uint64_t*)scratch.p, n_chunks, d_total);
    if (n_chunks) hipLaunchKernelGGL(k_fx_scan_add, dim3(n_chunks), dim3(1024), 0, c->stream, a, n, (const uint64_t*)scratch.p);
    GKC_HIP(c, hipGetLastError());
    return GKC_OK;
}

// ---------------------------------------------------------------------------------------------- text access
__device__ __forceinline__ void fx_load16(const uint8_t* text, uint64_t g0, uint64_t n, uint32_t (&dw)[4])
{
    if (g0 + 16 <= n) { const uint4 v = *reinterpret_cast<const uint4*>(text + g0); dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w; }
    else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { const uint64_t g = g0 + 4 * q + b; x |= (uint32_t)(g < n ? text[g] : 0u) << (8 * b); }   // padding is 0, never '\n'
            dw[q] = x;
        }
    }
}
__device__ __forceinline__ uint32_t fx_nl_mask(const uint32_t (&dw)[4])            // bit b set iff byte b of the 16 is '\n'
{
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t x = dw[q] ^ 0x0A0A0A0Au;
        uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;       // 0x80 where the byte is zero
        z >>= 7; z = (z | (z >> 7) | (z >> 14) | (z >> 21)) & 0xFu;
        m |= z << (4 * q);
    }
    return m;
}

// pass 1: newlines per tile
__global__ __launch_bounds__(FX_THREADS) void k_fx_count_nl(const uint8_t* __restrict__ text, uint64_t n, uint64_t* __restrict__ tile_nl)
{
    __shared__ uint32_t s_w[FX_THREADS / 64];
    uint32_t dw[4]; fx_load16(text, (uint64_t)blockIdx.x * FX_TILE + (uint64_t)threadIdx.x * FX_PER, n, dw);
    uint32_t cnt = __popc(fx_nl_mask(dw));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < FX_THREADS / 64; w++) t += s_w[w]; tile_nl[blockIdx.x] = t; }
}

// exclusive count of newlines before this thread's 16 bytes, inside the tile
__device__ __forceinline__ uint32_t fx_wg_excl(uint32_t v, uint32_t* s_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t p = 0;
    for (int w = 0; w < wave; w++) p += s_w[w];
    return p + x - v;
}

// pass 2: line_start[j + 1] = position after the j-th newline
__global__ __launch_bounds__(FX_THREADS) void k_fx_line_starts(const uint8_t* __restrict__ text, uint64_t n, const uint64_t* __restrict__ tile_line0,
                                                                uint64_t* __restrict__ line_start)
{
    __shared__ uint32_t s_w[FX_THREADS / 64];
    const uint64_t g0 = (uint64_t)blockIdx.x * FX_TILE + (uint64_t)threadIdx.x * FX_PER;
    uint32_t dw[4]; fx_load16(text, g0, n, dw);
    uint32_t m = fx_nl_mask(dw);
    uint64_t j = tile_line0[blockIdx.x] + fx_wg_excl(__popc(m), s_w);
    while (m) { const int b = __ffs((int)m) - 1; m &= m - 1; line_start[++j] = g0 + b + 1; }
}

// ---------------------------------------------------------------------------------------------- pass 3: lines
enum : uint8_t { FXK_SKIP = 0, FXK_HEADER = 1, FXK_SEQ = 2, FXK_STRIP = 4 };
struct FxInfo {                 // device-side results of the line passes
    unsigned long long first_header;      // index of the first line starting with '>' or '@' (~0 if none)
    unsigned long long last_header;       // index of the last such line (FASTA chunking)
    uint32_t error;                        // 0, or the first format violation found (FX_E_*)
    uint32_t pad;
    unsigned long long error_line;
};
enum : uint32_t { FX_E_PLUS_IN_FASTA = 1, FX_E_FASTQ_HEADER = 2, FX_E_FASTQ_SEQ = 3, FX_E_FASTQ_PLUS = 4, FX_E_FASTQ_QUAL = 5, FX_E_PREHEADER = 6 };

__device__ __forceinline__ void fx_error(FxInfo* info, uint32_t code, uint64_t line)
{
    if (atomicCAS(&info->error, 0u, code) == 0u) info->error_line = line;
}
__device__ __forceinline__ bool fx_is_hdr(uint32_t ch) { return ch == '>' || ch == '@'; }
// content length of line j (without its '\n'); line_start[n_lines] = n + 1 for an unterminated last line
__device__ __forceinline__ uint64_t fx_len(const uint64_t* line_start, uint64_t j) { return line_start[j + 1] - line_start[j] - 1; }

__global__ void k_fx_find_headers(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_start, uint64_t n_lines, FxInfo* info)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_lines) return;
    if (fx_len(line_start, j) > 0 && fx_is_hdr(text[line_start[j]])) { atomicMin(&info->first_header, (unsigned long long)j); atomicMax(&info->last_header, (unsigned long long)j); }
}

// length of line j after the reference's trailing-'\r' rule, for a line that is the ONLY sequence / quality line of its record
__device__ __forceinline__ uint64_t fx_len_single(const uint8_t* text, const uint64_t* line_start, uint64_t j)
{
    const uint64_t len = fx_len(line_start, j);
    return (len > 1 && text[line_start[j] + len - 1] == '\r') ? len - 1 : len;
}

// kinds of the lines [0, n_used); fastq != 0: strict 4-line records from line h0
__global__ void k_fx_classify(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_start, uint64_t n_lines, uint64_t n_used, uint64_t h0,
                              int fastq, uint64_t n_text, uint8_t* __restrict__ kind, uint64_t* __restrict__ is_hdr, FxInfo* info)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_lines) return;
    uint8_t kd = FXK_SKIP;
    if (j < n_used) {
        const uint64_t ls = line_start[j], len = fx_len(line_start, j);
        const uint32_t fc = len ? text[ls] : '\n';
        if (j < h0) {                                              // before the first header: skipped character by character by the reference
            for (uint64_t i = 0; i < len; i++) if (fx_is_hdr(text[ls + i])) { fx_error(info, FX_E_PREHEADER, j); break; }
        } else if (!fastq) {
            if (fx_is_hdr(fc)) kd = FXK_HEADER;
            else if (fc == '+') fx_error(info, FX_E_PLUS_IN_FASTA, j);
            else if (len) {
                kd = FXK_SEQ;
                if (text[ls + len - 1] == '\r') {
                    bool strip = len > 1;
                    if (!strip) {                                  // a line that is exactly "\r": dropped unless it is the first non-empty line of its record
                        for (uint64_t q = j; q-- > h0;) {
                            const uint64_t l2 = fx_len(line_start, q);
                            if (l2 && fx_is_hdr(text[line_start[q]])) break;
                            if (l2) { strip = true; break; }
                        }
                    }
                    if (strip) kd |= FXK_STRIP;
                }
            }
        } else {
            const uint32_t ph = (uint32_t)((j - h0) & 3u);
            if (ph == 0) { if (fx_is_hdr(fc)) kd = FXK_HEADER; else fx_error(info, FX_E_FASTQ_HEADER, j); }
            else if (ph == 1) {
                if (fx_is_hdr(fc) || fc == '+') fx_error(info, FX_E_FASTQ_SEQ, j);
                else if (len) { kd = FXK_SEQ; if (len > 1 && text[ls + len - 1] == '\r') kd |= FXK_STRIP; }
            } else if (ph == 2) { if (fc != '+') fx_error(info, FX_E_FASTQ_PLUS, j); }
            else if (fx_len_single(text, line_start, j) < fx_len_single(text, line_start, j - 2)) fx_error(info, FX_E_FASTQ_QUAL, j);
        }
    }
    // a lone '>' / '@' as the very last byte of the text starts no record: the reference finds nothing to read after it (BankFasta.cpp:505)
    if (kd == FXK_HEADER && line_start[j] + 1 == n_text) kd = FXK_SKIP;
    kind[j] = kd; is_hdr[j] = (kd == FXK_HEADER);
}

// ---------------------------------------------------------------------------------------------- passes 4 / 5: keep
// keep(byte i of line j) = kind[j] is SEQ, the byte is not the '\n', and it is not the dropped trailing '\r'
template <bool WRITE>
__global__ __launch_bounds__(FX_THREADS) void k_fx_keep(const uint8_t* __restrict__ text, uint64_t n, const uint64_t* __restrict__ tile_line0,
                                                         const uint64_t* __restrict__ line_start, const uint8_t* __restrict__ kind, uint64_t n_lines,
                                                         uint64_t* __restrict__ tile_keep /* COUNT: out; WRITE: exclusive prefix */,
                                                         const uint64_t* __restrict__ rec_id, uint8_t* __restrict__ bases, uint64_t* __restrict__ offsets)
{
    __shared__ uint32_t s_w[FX_THREADS / 64];
    __shared__ uint8_t s_out[FX_TILE];
    const uint64_t g0 = (uint64_t)blockIdx.x * FX_TILE + (uint64_t)threadIdx.x * FX_PER;
    uint32_t dw[4]; fx_load16(text, g0, n, dw);
    const uint32_t nl = fx_nl_mask(dw);
    uint64_t j = tile_line0[blockIdx.x] + fx_wg_excl(__popc(nl), s_w);
    __syncthreads();                                               // s_w is reused below
    // walk the 16 bytes; the line only changes after a newline
    uint32_t keepm = 0, hdrm = 0;
    uint64_t ls = 0, le = 0; uint8_t kd = FXK_SKIP; bool have = false;
#pragma unroll
    for (int b = 0; b < FX_PER; b++) {
        const uint64_t i = g0 + b;
        if (i < n) {
            if (!have) { if (j < n_lines) { ls = line_start[j]; le = line_start[j + 1] - 1; kd = kind[j]; } else { kd = FXK_SKIP; ls = ~0ULL; le = ~0ULL; } have = true; }
            const bool k = (kd & FXK_SEQ) && i < le && !((kd & FXK_STRIP) && i + 1 == le);
            keepm |= (uint32_t)k << b;
            hdrm |= (uint32_t)(kd == FXK_HEADER && i == ls) << b;
            if ((nl >> b) & 1) { j++; have = false; }
        }
    }
    const uint32_t cnt = __popc(keepm);
    const uint32_t rank0 = fx_wg_excl(cnt, s_w);
    if (!WRITE) {
        __syncthreads();
        if (threadIdx.x == FX_THREADS - 1) tile_keep[blockIdx.x] = rank0 + cnt;
        return;
    }
    const uint64_t out0 = tile_keep[blockIdx.x];
    // offsets[record] = number of kept bytes before the header line (a header line keeps nothing itself)
    {
        uint32_t r = rank0;
        uint64_t jl = j - __popc(nl);                              // line index of this thread's first byte
#pragma unroll
        for (int b = 0; b < FX_PER; b++) {
            if ((hdrm >> b) & 1) offsets[rec_id[jl]] = out0 + r;
            if ((keepm >> b) & 1) { s_out[r++] = (uint8_t)((dw[b >> 2] >> (8 * (b & 3))) & 255u); }
            if ((nl >> b) & 1) jl++;
        }
    }
    __syncthreads();
    __shared__ uint32_t s_total;
    if (threadIdx.x == FX_THREADS - 1) s_total = rank0 + cnt;
    __syncthreads();
    const uint32_t total = s_total;
    for (uint32_t q = threadIdx.x; q < total; q += FX_THREADS) bases[out0 + q] = s_out[q];
}

}   // namespace

// ------------------------------------------------------------------------------------------------ entry points
static const char* fx_error_text(uint32_t e)
{
    switch (e) {
        case FX_E_PLUS_IN_FASTA: return "a sequence line starts with '+' in a FASTA file (the reference would read a quality block there)";
        case FX_E_FASTQ_HEADER:  return "FASTQ: a record does not start with '@' / '>' (blank lines or multi-line records are not supported on the device)";
        case FX_E_FASTQ_SEQ:     return "FASTQ: the sequence line starts with '>', '@' or '+'";
        case FX_E_FASTQ_PLUS:    return "FASTQ: third line of a record does not start with '+' (multi-line FASTQ is not supported on the device)";
        case FX_E_FASTQ_QUAL:    return "FASTQ: quality shorter than its sequence (the reference would keep reading the next lines as quality)";
        case FX_E_PREHEADER:     return "'>' or '@' inside the lines before the first header line";
        default: return "unknown";
    }
}

int gkc_fastx_parse_device(gkc_ctx* c, const char* d_text, uint64_t n, int final_chunk, char** d_bases, uint64_t** d_offsets,
                           uint64_t* n_reads, uint64_t* n_bases, uint64_t* consumed)
{
    if (!c || !d_bases || !d_offsets || !n_reads || !n_bases || !consumed) return GKC_ERR_ARG;
    *d_bases = nullptr; *d_offsets = nullptr; *n_reads = 0; *n_bases = 0; *consumed = 0;
    GKC_HIP(c, hipSetDevice(c->device));
    if (n >= (1ULL << 36)) GKC_FAIL(c, GKC_ERR_ARG, "a text chunk is limited to 2^36 bytes");
    const uint8_t* text = (const uint8_t*)d_text;
    auto finish_empty = [&]() -> int {
        void* o = c->dalloc(8); void* b = o ? c->dalloc(64) : nullptr;           // (the context's caching allocator: a chunk per call would otherwise cost a hipMalloc + hipFree of ~10 ms)
        if (!o || !b) { if (o) c->dfree(o); return GKC_ERR_NOMEM; }
        GKC_HIP(c, hipMemsetAsync(o, 0, 8, c->stream)); GKC_HIP(c, hipStreamSynchronize(c->stream));
        *d_bases = (char*)b; *d_offsets = (uint64_t*)o; *consumed = final_chunk ? n : 0;
        return GKC_OK;
    };
    if (n == 0) return finish_empty();

    const uint64_t n_tiles = (n + FX_TILE - 1) / FX_TILE;
    DevBuf tile_line0, tile_keep, scratch, d_info, d_tot, line_start, kind, rec_id;
    struct Guard { std::vector<DevBuf*> v; ~Guard() { for (DevBuf* b : v) b->release(); } } guard;
    guard.v = { &tile_line0, &tile_keep, &scratch, &d_info, &d_tot, &line_start, &kind, &rec_id };
    GKC_TRY(c->ensure(tile_line0, (size_t)n_tiles * 8)); GKC_TRY(c->ensure(tile_keep, (size_t)n_tiles * 8));
    GKC_TRY(c->ensure(d_info, sizeof(FxInfo))); GKC_TRY(c->ensure(d_tot, 64));

    // 1 newlines per tile, scan
    hipLaunchKernelGGL(k_fx_count_nl, dim3((unsigned)n_tiles), dim3(FX_THREADS), 0, c->stream, text, n, (uint64_t*)tile_line0.p);
    GKC_TRY(fx_scan(c, (uint64_t*)tile_line0.p, n_tiles, (uint64_t*)d_tot.p, scratch));
    uint64_t total_nl = 0; uint8_t last_byte = 0;
    GKC_HIP(c, hipMemcpyAsync(&total_nl, d_tot.p, 8, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipMemcpyAsync(&last_byte, text + n - 1, 1, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    const bool open_tail = last_byte != '\n';
    const uint64_t n_lines = total_nl + (open_tail ? 1 : 0);
    // 2 line table
    GKC_TRY(c->ensure(line_start, (size_t)(n_lines + 2) * 8));
    GKC_HIP(c, hipMemsetAsync(line_start.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_fx_line_starts, dim3((unsigned)n_tiles), dim3(FX_THREADS), 0, c->stream, text, n, (const uint64_t*)tile_line0.p, (uint64_t*)line_start.p);
    if (open_tail) { const uint64_t v = n + 1; GKC_HIP(c, hipMemcpyAsync((uint64_t*)line_start.p + n_lines, &v, 8, hipMemcpyHostToDevice, c->stream)); GKC_HIP(c, hipStreamSynchronize(c->stream)); }
    // 3 first / last header, mode
    FxInfo info{}; info.first_header = ~0ULL; info.last_header = 0;
    GKC_HIP(c, hipMemcpyAsync(d_info.p, &info, sizeof(info), hipMemcpyHostToDevice, c->stream));
    const unsigned lgrid = (unsigned)((n_lines + 255) / 256);
    hipLaunchKernelGGL(k_fx_find_headers, dim3(lgrid), dim3(256), 0, c->stream, text, (const uint64_t*)line_start.p, n_lines, (FxInfo*)d_info.p);
    GKC_HIP(c, hipMemcpyAsync(&info, d_info.p, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    if (info.first_header == ~0ULL) {                              // no header at all: nothing to emit (the reference skips everything)
        // a non-final chunk without any header cannot be cut: hand it all back
        return finish_empty();
    }
    const uint64_t h0 = info.first_header;
    int fastq = 0;
    if (h0 + 2 < n_lines) {
        uint64_t ls3[4]; GKC_HIP(c, hipMemcpy(ls3, (uint64_t*)line_start.p + h0 + 1, 3 * 8, hipMemcpyDeviceToHost));
        uint8_t c1 = '\n', c2 = '\n';
        if (ls3[1] - ls3[0] > 1) GKC_HIP(c, hipMemcpy(&c1, text + ls3[0], 1, hipMemcpyDeviceToHost));
        if (ls3[2] - ls3[1] > 1) GKC_HIP(c, hipMemcpy(&c2, text + ls3[1], 1, hipMemcpyDeviceToHost));
        fastq = (c2 == '+' && c1 != '>' && c1 != '@' && c1 != '+');
    }
    // lines that belong to complete records of this chunk
    uint64_t n_used = n_lines, cons = n;
    if (!final_chunk) {
        std::vector<uint64_t> one(1);
        if (fastq) { const uint64_t recs = total_nl > h0 ? (total_nl - h0) / 4 : 0; n_used = h0 + 4 * recs; }
        else n_used = info.last_header;                            // the last record may continue in the next chunk
        GKC_HIP(c, hipMemcpy(one.data(), (uint64_t*)line_start.p + n_used, 8, hipMemcpyDeviceToHost));
        cons = n_used < n_lines ? one[0] : n;
    }
    // classify
    GKC_TRY(c->ensure(kind, (size_t)n_lines + 8)); GKC_TRY(c->ensure(rec_id, (size_t)(n_lines + 1) * 8));
    hipLaunchKernelGGL(k_fx_classify, dim3(lgrid), dim3(256), 0, c->stream, text, (const uint64_t*)line_start.p, n_lines, n_used, h0, fastq, n,
                       (uint8_t*)kind.p, (uint64_t*)rec_id.p, (FxInfo*)d_info.p);
    GKC_TRY(fx_scan(c, (uint64_t*)rec_id.p, n_lines, (uint64_t*)d_tot.p, scratch));
    // 4 kept bytes per tile
    hipLaunchKernelGGL((k_fx_keep<false>), dim3((unsigned)n_tiles), dim3(FX_THREADS), 0, c->stream, text, n, (const uint64_t*)tile_line0.p, (const uint64_t*)line_start.p,
                       (const uint8_t*)kind.p, n_lines, (uint64_t*)tile_keep.p, (const uint64_t*)nullptr, (uint8_t*)nullptr, (uint64_t*)nullptr);
    GKC_TRY(fx_scan(c, (uint64_t*)tile_keep.p, n_tiles, (uint64_t*)d_tot.p + 1, scratch));
    uint64_t tots[2] = {0, 0};
    GKC_HIP(c, hipMemcpyAsync(tots, d_tot.p, 16, hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipMemcpyAsync(&info, d_info.p, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    GKC_HIP(c, hipStreamSynchronize(c->stream));
    if (info.error) GKC_FAIL(c, GKC_ERR_FORMAT, "FASTA/FASTQ text not parseable on the device, line %llu: %s", (unsigned long long)info.error_line + 1, fx_error_text(info.error));
    const uint64_t nr = tots[0], nb = tots[1];
    // 5 compaction
    void* b = c->dalloc((size_t)nb + 64);
    if (!b) return GKC_ERR_NOMEM;
    void* o = c->dalloc((size_t)(nr + 1) * 8);
    if (!o) { c->dfree(b); return GKC_ERR_NOMEM; }
    hipLaunchKernelGGL((k_fx_keep<true>), dim3((unsigned)n_tiles), dim3(FX_THREADS), 0, c->stream, text, n, (const uint64_t*)tile_line0.p, (const uint64_t*)line_start.p,
                       (const uint8_t*)kind.p, n_lines, (uint64_t*)tile_keep.p, (const uint64_t*)rec_id.p, (uint8_t*)b, (uint64_t*)o);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync((uint64_t*)o + nr, &nb, 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->dfree(b); c->dfree(o); GKC_FAIL(c, GKC_ERR_HIP, "fastx compaction failed: %s", hipGetErrorString(e)); }
    *d_bases = (char*)b; *d_offsets = (uint64_t*)o; *n_reads = nr; *n_bases = nb; *consumed = cons;
    return GKC_OK;
}

int gkc_push_fastx(gkc_ctx* c, const char* text, uint64_t n, int final_chunk, uint64_t* consumed)
{
    if (!c || !consumed) return GKC_ERR_ARG;
    if (!c->in_pass) GKC_FAIL(c, GKC_ERR_ARG, "gkc_begin_pass must be called first");
    GKC_HIP(c, hipSetDevice(c->device));
    DevBuf dt;
    GKC_TRY(c->ensure(dt, (size_t)n + 64));
    if (n) { hipError_t e = hipMemcpyAsync(dt.p, text, (size_t)n, hipMemcpyHostToDevice, c->stream); if (e != hipSuccess) { dt.release(); GKC_FAIL(c, GKC_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e)); } }
    char* db = nullptr; uint64_t* dof = nullptr; uint64_t nr = 0, nb = 0;
    int rc = gkc_fastx_parse_device(c, (const char*)dt.p, n, final_chunk, &db, &dof, &nr, &nb, consumed);
    if (rc == GKC_OK && nr) rc = gkc_push_reads_device(c, db, dof, nr, nb);
    (void)hipStreamSynchronize(c->stream);
    if (db) c->dfree(db);
    if (dof) c->dfree(dof);
    dt.release();
    return rc;
}
