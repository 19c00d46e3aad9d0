// c2_api_fastq.hip -- host side of the C ABI declared in include/crispresso2_amd.h: FASTQ framing + exact de-duplication on the device (launch-only; crispresso2_amd/fastq_device.py drives them).
// Marshals the caller's inputs into the kernels' tables, owns the device buffers of a context, picks launch geometry and
// launches.  Nothing here computes an alignment or a classification on the CPU.
#include "c2_ctx.h"
#include "c2_k_fastq.hip"

extern "C" {

// ---- FASTQ framing + de-duplication on the device (launch-only; crispresso2_amd/fastq_device.py drives them) ----
static_assert(C2_FQ_TILE == C2_FQ_TILE_BYTES, "header and kernels disagree on the framing tile");
int c2_fq_count_device(c2_ctx* ctx, const uint8_t* d_text, uint64_t lo, uint64_t hi, uint32_t* d_tile_newlines, uint32_t* d_tile_empty,
                       uint32_t* d_flags, void* hip_stream) {
    if (!ctx || !d_text || !d_tile_newlines || !d_tile_empty || !d_flags || hi < lo || (lo % 16u)) { if (ctx) ctx->err = "bad argument"; return C2_E_INVALID; }
    if (hi == lo) return 0;
    const uint64_t tiles = (hi - lo + C2_FQ_TILE - 1) / C2_FQ_TILE;
    if (tiles > 0x7fffffffull) { ctx->err = "range too large for one launch"; return C2_E_TOO_LARGE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_fq_frame_args A{};
    A.text = d_text; A.lo = lo; A.hi = hi; A.tile_newlines = d_tile_newlines; A.tile_empty = d_tile_empty; A.flags = d_flags;
    hipLaunchKernelGGL(c2_fq_count_kernel, dim3((unsigned)tiles), dim3(256), C2_FQ_LDS_BYTES, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_fq_lines_device(c2_ctx* ctx, const uint8_t* d_text, uint64_t lo, uint64_t hi, const uint64_t* d_tile_base, uint64_t* d_seq_start,
                       uint64_t* d_seq_end, uint64_t n_records_cap, void* hip_stream) {
    if (!ctx || !d_text || !d_tile_base || !d_seq_start || !d_seq_end || hi < lo || (lo % 16u)) { if (ctx) ctx->err = "bad argument"; return C2_E_INVALID; }
    if (hi == lo) return 0;
    const uint64_t tiles = (hi - lo + C2_FQ_TILE - 1) / C2_FQ_TILE;
    if (tiles > 0x7fffffffull) { ctx->err = "range too large for one launch"; return C2_E_TOO_LARGE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_fq_frame_args A{};
    A.text = d_text; A.lo = lo; A.hi = hi; A.tile_base = d_tile_base; A.seq_start = d_seq_start; A.seq_end = d_seq_end; A.n_records_cap = n_records_cap;
    hipLaunchKernelGGL(c2_fq_lines_kernel, dim3((unsigned)tiles), dim3(256), C2_FQ_LDS_BYTES, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_fq_lines4_device(c2_ctx* ctx, const uint8_t* d_text, uint64_t lo, uint64_t hi, const uint64_t* d_tile_base, uint64_t* d_seq_start,
                        uint64_t* d_seq_end, uint64_t* d_qual_start, uint64_t* d_qual_end, uint64_t n_records_cap, void* hip_stream) {
    if (!ctx || !d_text || !d_tile_base || !d_seq_start || !d_seq_end || !d_qual_start || !d_qual_end || hi < lo || (lo % 16u)) { if (ctx) ctx->err = "bad argument"; return C2_E_INVALID; }
    if (hi == lo) return 0;
    const uint64_t tiles = (hi - lo + C2_FQ_TILE - 1) / C2_FQ_TILE;
    if (tiles > 0x7fffffffull) { ctx->err = "range too large for one launch"; return C2_E_TOO_LARGE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_fq_frame_args A{};
    A.text = d_text; A.lo = lo; A.hi = hi; A.tile_base = d_tile_base; A.seq_start = d_seq_start; A.seq_end = d_seq_end; A.n_records_cap = n_records_cap;
    A.qual_start = d_qual_start; A.qual_end = d_qual_end;
    hipLaunchKernelGGL(c2_fq_lines_kernel, dim3((unsigned)tiles), dim3(256), C2_FQ_LDS_BYTES, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

static bool c2_fq_pair_fill(c2_fq_pair_args& A, const uint8_t* d_text1, const uint8_t* d_text2, const uint64_t* const* d_lines1, const uint64_t* const* d_lines2,
                            uint64_t n, uint64_t* d_s1, uint64_t* d_q1, uint64_t* d_s2, uint64_t* d_q2, uint32_t* d_flags) {
    if (!d_text1 || !d_text2 || !d_lines1 || !d_lines2 || !d_s1 || !d_q1 || !d_s2 || !d_q2 || !d_flags) return false;
    for (int q = 0; q < 4; ++q) if (!d_lines1[q] || !d_lines2[q]) return false;
    A.text1 = d_text1; A.text2 = d_text2;
    A.seq_start1 = d_lines1[0]; A.seq_end1 = d_lines1[1]; A.qual_start1 = d_lines1[2]; A.qual_end1 = d_lines1[3];
    A.seq_start2 = d_lines2[0]; A.seq_end2 = d_lines2[1]; A.qual_start2 = d_lines2[2]; A.qual_end2 = d_lines2[3];
    A.n = n;
    A.s1 = (unsigned long long*)d_s1; A.q1 = (unsigned long long*)d_q1; A.s2 = (unsigned long long*)d_s2; A.q2 = (unsigned long long*)d_q2;
    A.flags = d_flags;
    return true;
}

int c2_fq_pair_lengths_device(c2_ctx* ctx, const uint8_t* d_text1, const uint8_t* d_text2, const uint64_t* const* d_lines1, const uint64_t* const* d_lines2,
                              uint64_t n, uint64_t* d_s1, uint64_t* d_q1, uint64_t* d_s2, uint64_t* d_q2, int64_t* d_key_len, int64_t* d_qual_len,
                              uint32_t* d_flags, void* hip_stream) {
    c2_fq_pair_args A{};
    if (!ctx || !c2_fq_pair_fill(A, d_text1, d_text2, d_lines1, d_lines2, n, d_s1, d_q1, d_s2, d_q2, d_flags) || !d_key_len || !d_qual_len) {
        if (ctx) ctx->err = "bad argument";
        return C2_E_INVALID;
    }
    if (n == 0) return 0;
    if ((n + 255) / 256 > 0x7fffffffull) { ctx->err = "too many records for one launch"; return C2_E_TOO_LARGE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    A.key_len = d_key_len; A.qual_len = d_qual_len;
    hipLaunchKernelGGL(c2_fq_pair_lengths_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_fq_pair_write_device(c2_ctx* ctx, const uint8_t* d_text1, const uint8_t* d_text2, uint64_t n, const uint64_t* d_s1, const uint64_t* d_q1,
                            const uint64_t* d_s2, const uint64_t* d_q2, const int64_t* d_key_off, const int64_t* d_qual_off, uint8_t* d_key_out,
                            uint8_t* d_qual_out, uint32_t* d_flags, void* hip_stream) {
    if (!ctx || !d_text1 || !d_text2 || !d_s1 || !d_q1 || !d_s2 || !d_q2 || !d_key_off || !d_qual_off || !d_key_out || !d_qual_out || !d_flags) {
        if (ctx) ctx->err = "bad argument";
        return C2_E_INVALID;
    }
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_fq_pair_args A{};
    A.text1 = d_text1; A.text2 = d_text2; A.n = n;
    A.s1 = (unsigned long long*)d_s1; A.q1 = (unsigned long long*)d_q1; A.s2 = (unsigned long long*)d_s2; A.q2 = (unsigned long long*)d_q2;
    A.key_off = d_key_off; A.qual_off = d_qual_off; A.key_out = d_key_out; A.qual_out = d_qual_out; A.flags = d_flags;
    const uint64_t wgs = std::min<uint64_t>((n + 3) / 4, (uint64_t)ctx->prop.multiProcessorCount * 16u);
    hipLaunchKernelGGL(c2_fq_pair_write_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_fq_dedup_device(c2_ctx* ctx, const uint8_t* d_text, const uint64_t* d_seq_start, const uint64_t* d_seq_end, const uint64_t* d_range,
                       uint64_t n_records_cap, uint64_t* d_slots, uint64_t n_slots, uint32_t* d_count, uint32_t* d_first,
                       uint32_t* d_slot_of, uint64_t* d_rinfo, uint32_t* d_flags, uint32_t* d_stats, void* hip_stream) {
    if (!ctx || !d_text || !d_seq_start || !d_seq_end || !d_range || !d_slots || !d_count || !d_first || !d_slot_of || !d_rinfo || !d_flags ||
        !d_stats || n_slots < 2 || (n_slots & (n_slots - 1)) || n_slots > 0xffffffffull) { if (ctx) ctx->err = "bad argument"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_fq_dedup_args A{};
    A.text = d_text; A.seq_start = d_seq_start; A.seq_end = d_seq_end; A.range = d_range; A.n_records_cap = n_records_cap;
    A.slots = (unsigned long long*)d_slots; A.mask = n_slots - 1; A.count = d_count; A.first = d_first; A.slot_of = d_slot_of;
    A.rinfo = (unsigned long long*)d_rinfo; A.flags = d_flags; A.stats = d_stats;
    // grid-stride over the records (how many is only known on the device): enough wavefronts to cover the latency of the table probes
    hipLaunchKernelGGL(c2_fq_dedup_kernel, dim3((unsigned)ctx->prop.multiProcessorCount * 8u), dim3(256), C2_FQ_DEDUP_LDS_BYTES, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_fq_gather_device(c2_ctx* ctx, const uint8_t* d_text, const uint64_t* d_info, const int64_t* d_records, const int64_t* d_out_offsets,
                        uint8_t* d_out, uint64_t n, void* hip_stream) {
    if (!ctx || !d_text || !d_info || !d_out_offsets || !d_out) { if (ctx) ctx->err = "bad argument"; return C2_E_INVALID; }
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_fq_gather_args A{};
    A.text = d_text; A.info = (const unsigned long long*)d_info; A.records = d_records; A.out_offsets = d_out_offsets; A.out = d_out; A.n = n;
    const uint64_t wgs = std::min<uint64_t>((n + 3) / 4, (uint64_t)ctx->prop.multiProcessorCount * 16u);
    hipLaunchKernelGGL(c2_fq_gather_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int c2_fq_rc_partner_device(c2_ctx* ctx, const uint8_t* d_text, const uint64_t* d_info, const int64_t* d_records, uint64_t n,
                            const uint64_t* d_slots, uint64_t n_slots, int32_t* d_partner_slot, void* hip_stream) {
    if (!ctx || !d_text || !d_info || !d_records || !d_slots || !d_partner_slot || n_slots < 2 || (n_slots & (n_slots - 1)) || n_slots > 0x7fffffffull) {
        if (ctx) ctx->err = "bad argument";
        return C2_E_INVALID;
    }
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c2_fq_rc_args A{};
    A.text = d_text; A.info = (const unsigned long long*)d_info; A.records = d_records; A.n = n; A.slots = (const unsigned long long*)d_slots;
    A.mask = n_slots - 1; A.partner_slot = d_partner_slot;
    const uint64_t wgs = std::min<uint64_t>((n + 3) / 4, (uint64_t)ctx->prop.multiProcessorCount * 16u);
    hipLaunchKernelGGL(c2_fq_rc_partner_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)hip_stream, A);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
