// c2_api_alleles.hip -- host side of the C ABI declared in include/crispresso2_amd.h: the allele frequency table (c2_allele_table_*).
// The orchestration lives in c2_alleles_host.h (shared with the test-only wave emulator); this file is its HIP backend: device and
// page-locked memory, the kernels of c2_k_alleles.hip, rocPRIM's device merge sort (with the table's order as its comparator) and scan.
#include "c2_ctx.h"
#include <cstring>
#include <memory>
#include <rocprim/rocprim.hpp>
#include "c2_k_alleles.hip"
#include "c2_alleles_host.h"

namespace {

struct ToU64 { __device__ uint64_t operator()(const uint32_t x) const { return (uint64_t)x; } };

__global__ __launch_bounds__(256) void c2_iota_kernel(uint32_t* out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

struct HipBackend {
    c2_ctx* ctx;
    hipStream_t s;
    std::string error;

    bool ok(hipError_t e, const char* what) { if (e == hipSuccess) return true; error = std::string(what) + ": " + hipGetErrorString(e); return false; }
    void* dalloc(size_t n) { void* p = nullptr; return ok(hipMalloc(&p, std::max<size_t>(n, 256)), "hipMalloc") ? p : nullptr; }
    void dfree(void* p) { if (p) (void)hipFree(p); }
    // page-locked staging: the context keeps its two output buffers (the host batch path's) -- pinning 2 x 64 MiB per call would cost more
    // than writing the first chunk
    int pin_slot = 0;
    void* halloc(size_t n) {
        if (pin_slot < 2) { const int k = pin_slot++; return ensure_pinned(ctx, ctx->pin_out[k], ctx->pin_out_cap[k], n) == 0 ? ctx->pin_out[k] : nullptr; }
        void* p = nullptr; return ok(hipHostMalloc(&p, n, hipHostMallocDefault), "hipHostMalloc") ? p : nullptr;
    }
    void hfree(void* p) { if (p && p != ctx->pin_out[0] && p != ctx->pin_out[1]) (void)hipHostFree(p); else if (p) pin_slot = 0; }
    bool h2d(void* d, const void* h, size_t n) { return n == 0 || (ok(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s), "hipMemcpy H2D") && ok(hipStreamSynchronize(s), "sync")); }
    bool d2h(void* h, const void* d, size_t n) { return n == 0 || (ok(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "hipMemcpy D2H") && ok(hipStreamSynchronize(s), "sync")); }
    bool zero(void* d, size_t n) { return ok(hipMemsetAsync(d, 0, n, s), "hipMemset"); }
    bool sync() { return ok(hipStreamSynchronize(s), "hipStreamSynchronize"); }
    static unsigned blocks(uint64_t n, unsigned per) { return (unsigned)std::max<uint64_t>(1, (n + per - 1) / per); }
    bool launched(const char* what) { return ok(hipGetLastError(), what); }

    bool jobs(const c2_allele_jobs_args& A) { hipLaunchKernelGGL(c2_allele_jobs_kernel, dim3(blocks(A.S.n_reads, 256)), dim3(256), 0, s, A); return launched("c2_allele_jobs_kernel"); }
    bool iota(uint32_t* out, uint64_t n) { hipLaunchKernelGGL(c2_iota_kernel, dim3(blocks(n, 256)), dim3(256), 0, s, out, n); return launched("c2_iota_kernel"); }
    bool reads(const c2_allele_row* rows, const uint32_t* order, uint64_t m, uint32_t* out) {
        hipLaunchKernelGGL(c2_allele_reads_kernel, dim3(blocks(m, 256)), dim3(256), 0, s, rows, order, m, out); return launched("c2_allele_reads_kernel"); }
    bool probe(const c2_allele_probe_args& A) { hipLaunchKernelGGL(c2_allele_probe_kernel, dim3(blocks(A.m, 4)), dim3(256), 0, s, A); return launched("c2_allele_probe_kernel"); }
    bool lengths(const c2_allele_text_args& A) { hipLaunchKernelGGL(c2_allele_lengths_kernel, dim3(blocks(A.m, 256)), dim3(256), 0, s, A); return launched("c2_allele_lengths_kernel"); }
    bool emit(const c2_allele_text_args& A) { hipLaunchKernelGGL(c2_allele_emit_kernel, dim3(blocks(A.q1 - A.q0, 4)), dim3(256), 512, s, A); return launched("c2_allele_emit_kernel"); }
    bool fetch(const c2_allele_fetch_args& A) { hipLaunchKernelGGL(c2_allele_fetch_kernel, dim3(blocks(A.m, 4)), dim3(256), 0, s, A); return launched("c2_allele_fetch_kernel"); }
    bool window(const c2_allele_window_args& A) {
        hipLaunchKernelGGL(c2_allele_window_kernel, dim3(blocks(A.m, A.sub_index ? 4 : 256)), dim3(256), 0, s, A); return launched("c2_allele_window_kernel"); }
    bool group(const c2_allele_group_args& A) { hipLaunchKernelGGL(c2_allele_group_kernel, dim3(blocks(A.ms, 256)), dim3(256), 0, s, A); return launched("c2_allele_group_kernel"); }

    bool scan(const uint32_t* d_in, uint64_t* d_out, uint64_t n) {
        size_t need = 0;
        auto it = rocprim::make_transform_iterator(d_in, ToU64());
        if (!ok(rocprim::exclusive_scan(nullptr, need, it, d_out, (uint64_t)0, n, rocprim::plus<uint64_t>(), s), "rocprim::exclusive_scan")) return false;
        void* tmp = dalloc(need);
        if (!tmp) return false;
        const bool r = ok(rocprim::exclusive_scan(tmp, need, it, d_out, (uint64_t)0, n, rocprim::plus<uint64_t>(), s), "rocprim::exclusive_scan") && sync();
        dfree(tmp);
        return r;
    }
    template <class Less>
    bool sort_with(Less less, uint32_t* d_in, uint32_t* d_out, uint64_t n) {
        size_t need = 0;
        if (!ok(rocprim::merge_sort(nullptr, need, d_in, d_out, n, less, s), "rocprim::merge_sort")) return false;
        void* tmp = dalloc(need);
        if (!tmp) return false;
        const bool r = ok(rocprim::merge_sort(tmp, need, d_in, d_out, n, less, s), "rocprim::merge_sort") && sync();
        dfree(tmp);
        return r;
    }
    bool sort_rows(c2_allele_row_less less, uint32_t* d_in, uint32_t* d_out, uint64_t n) { return sort_with(less, d_in, d_out, n); }
    bool sort_keys(c2_allele_key_less less, uint32_t* d_in, uint32_t* d_out, uint64_t n) { return sort_with(less, d_in, d_out, n); }
};

}  // namespace

struct c2_allele_table { c2a_table<HipBackend>* t; c2_ctx* ctx; };

extern "C" {

int c2_allele_table_build(c2_ctx* ctx, const c2_allele_src* src, c2_allele_table** out, void* hip_stream) {
    if (!ctx || !src || !out) { if (ctx) ctx->err = "NULL argument"; return C2_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HipBackend be{ctx, hip_stream ? (hipStream_t)hip_stream : ctx->stream, std::string()};
    c2a_table<HipBackend>* t = nullptr;
    const int rc = c2a_build(be, *src, &t, ctx->err);
    if (rc) return rc;
    *out = new c2_allele_table{t, ctx};
    return 0;
}

uint64_t c2_allele_table_rows(const c2_allele_table* t) { return t ? t->t->m : 0; }

int c2_allele_table_write(c2_allele_table* t, const char* path, const char* const* labels, int64_t n_total, const char* const* probes,
                          int32_t threads, uint64_t* bytes_written) {
    if (!t || !path) return C2_E_INVALID;
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    const int rc = c2a_write(t->t, path, labels, n_total, probes, threads, bytes_written);
    if (rc) t->ctx->err = t->t->err;
    return rc;
}

int c2_allele_table_write_zip(c2_allele_table* t, const char* zip_path, const char* member, const char* const* labels, int64_t n_total,
                              const char* const* probes, int32_t threads, int32_t level, uint64_t* text_bytes, uint64_t* zip_bytes) {
    if (!t || !zip_path || !member || !member[0]) return C2_E_INVALID;
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    const int rc = c2a_write(t->t, zip_path, labels, n_total, probes, threads, text_bytes, member, level, zip_bytes);
    if (rc) t->ctx->err = t->t->err;
    return rc;
}

int c2_allele_table_fetch(c2_allele_table* t, c2_allele_row* rows, uint8_t* aligned, uint8_t* reference, uint32_t stride) {
    if (!t) return C2_E_INVALID;
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    const int rc = c2a_fetch(t->t, rows, aligned, reference, stride);
    if (rc) t->ctx->err = t->t->err;
    return rc;
}

int c2_allele_table_around_cut_write(c2_allele_table* t, int32_t label, int32_t cut_point, int32_t ref_len, int32_t plot_window_size,
                                     int64_t n_total, const char* path, int32_t threads, uint64_t* n_groups) {
    if (!t || !path) return C2_E_INVALID;
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    const int rc = c2a_around_cut_write(t->t, label, cut_point, ref_len, plot_window_size, n_total, path, threads, n_groups);
    if (rc) t->ctx->err = t->t->err;
    return rc;
}

void c2_allele_table_free(c2_allele_table* t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    delete t->t;
    delete t;
}

int c2_format_float_repr(double v, char* out) { return out ? c2_py_float_repr(v, out) : 0; }

}  // extern "C"
