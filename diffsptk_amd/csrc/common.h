// Shared host/device helpers for the diffsptk_amd HIP library (gfx950 only).
#pragma once

// Translation units are built with the compiler's packed-float32 selection switched OFF (_lib.py: SOURCE_FLAGS): its own
// v_pk_*_f32 code pairs registers with extra moves, and it freely emits the forms with a set op_sel bit (a LOW result half
// reading a HIGH source half) -- the instruction class of every transient wrong result DESIGN.md 4 recorded, which no shipped
// kernel may execute (tests/test_host_cpu.py::test_no_crossed_packed_float32).  Switching the feature off also makes the
// assembler reject v_pk_*_f32 in inline assembly: a kernel that places packed instructions by hand carries this attribute,
// which switches the feature back on for that function only.
#if defined(__HIP_DEVICE_COMPILE__)
#define DSA_PK_TARGET __attribute__((target("packed-fp32-ops")))
#else
#define DSA_PK_TARGET
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/diffsptk_amd.h"

#define DSA_EXPORT extern "C" __attribute__((visibility("default")))

namespace dsa {

// ---- per-thread error / dispatch record -------------------------------------------------
inline char* err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}
inline const char*& kernel_name()
{
    static thread_local const char* name = "";
    return name;
}
inline int fail(int code, const char* fmt, const char* detail = "")
{
    snprintf(err_buf(), 512, fmt, detail);
    return code;
}
inline int check_launch(const char* name)
{
    kernel_name() = name;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(err_buf(), 512, "%s: launch failed: %s", name, hipGetErrorString(e));
        return DSA_ERR_LAUNCH;
    }
    return DSA_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): set once per device the process
// drives (`done` = bit mask of the devices that have it; a process may drive several GPUs).
inline bool ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done)
{
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    const uint64_t bit = known ? (uint64_t)1 << dev : 0;
    if (known && (done.load(std::memory_order_acquire) & bit)) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (known) done.fetch_or(bit, std::memory_order_release);
    return true;
}

// matrix-core "transform of the spectrum x (K x C) matrix" launcher of fbank.hip, shared with fftcep.hip
// (use_power: 0 sqrt x | 1 x | 2 log x;  post_mode 0: floor + glog, 1: scale with the first column halved,
//  3: first and last column halved;  needs float32, C <= 48, C < K <= 320)
// 24 x 24 Toeplitz-plus-Hankel systems, float32, 16 per wave in the quad layout (csrc/mcep_mfma.hip)
int thsolve_quad24_fwd(const void* p, const void* q, const void* r, int64_t F, void* g, hipStream_t st, int r_stride = 24,
                       int r_off = 0, const void* add = nullptr);
// the whole Newton step of mgcep (gamma != 0, fft_length 512, cep_order 24, float32) in one launch: csrc/mgcep_step_f16.h
int mgcep_step_solve_fwd(const void* x, const void* b1, int64_t F, double gamma, const void* images, void* b1_out, void* r_out, hipStream_t st,
                         void* pt_out = nullptr, void* qt_out = nullptr, int n_steps = 1, void* b1_prev_out = nullptr);
int mgcep_step_bwd_h(const void* x, const void* b1, const void* gpt, const void* gqt, const void* gr, int64_t F, double gamma,
                     const void* images, const void* gx_in, void* gx, void* gb1, hipStream_t st);
// the spectral half of a Newton step of the untuned mel-cepstral analysis on binary16-split chains: csrc/mcep_resid_f16.h
int64_t mcep_resid_h_images_bytes(int K, int M1);
int mcep_resid_h_prepare(const void* D, int ldd, const void* E, int lde, int K, int M1, void* images, hipStream_t st);
int mcep_resid_h_fwd(const void* logx, int64_t F, int K, const void* mc, int M1, const void* images, void* out, int ldo, hipStream_t st);
int64_t mcep_resid_bwd_images_bytes(int K, int M1);
int mcep_resid_bwd_prepare(const void* D, int ldd, const void* E, int lde, int K, int M1, void* images, hipStream_t st);
int mcep_glogx_h(const void* logx, int64_t F, int K, const void* mcs, int M1, const void* grts, int n_iter, const void* images, void* glogx,
                 hipStream_t st);
int mcep_resid_bwd_h(const void* logx, int64_t F, int K, const void* mc, int M1, const void* grt, const void* images, void* glogx, void* gmc,
                     hipStream_t st);
int mcep_big_newton(const void* logx, int64_t F, int K, const void* mc_in, int M1, const void* images, const void* av, int n_iter,
                    void* mc_out, hipStream_t st);   // csrc/mcep_mfma.hip (mcep_big_f16.h)
// orders 2 .. 55, float32, strided operands (csrc/thsolve_quad.hip)
int thsolve_quadn_fwd(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add, int64_t F,
                      int n, void* g, hipStream_t st);
int fbank_mfma_launch_ex(const void* x, int64_t F, int K, const void* H, int C, int ldh, double floor, double gamma,
                         int use_power, int post_mode, double post_scale, void* y, void* E, hipStream_t st, const char* name,
                         const void* W2 = nullptr, int Mo = 0, void* z = nullptr,   // optional second product z = y W2
                         int ldy = 0);   // row stride of y when it is a column slice of a wider matrix (0: C); post_mode 4: plain product

#define DSA_REQUIRE(cond, msg)                                              \
    do {                                                                    \
        if (!(cond)) return dsa::fail(DSA_ERR_INVALID_ARGUMENT, "%s", msg); \
    } while (0)

// ---- device helpers ------------------------------------------------------------------------
// F.pad index semantics of Frame (frame.py:134-137).  i indexes the UN-padded signal of
// length T; returns the source index, or -1 for a zero (constant mode).
__device__ __forceinline__ long pad_src_index(long i, long T, int mode)
{
    if (i >= 0 && i < T) return i;
    switch (mode) {
    case DSA_PAD_REFLECT: {
        if (T == 1) return 0;
        long period = 2 * (T - 1);
        long j = i % period;
        if (j < 0) j += period;
        return j < T ? j : period - j;
    }
    case DSA_PAD_REPLICATE: return i < 0 ? 0 : T - 1;
    case DSA_PAD_CIRCULAR: {
        long j = i % T;
        if (j < 0) j += T;
        return j;
    }
    default: return -1;
    }
}

template <typename T>
__device__ __forceinline__ T load_padded(const T* __restrict__ x, long i, long len, int mode)
{
    long j = pad_src_index(i, len, mode);
    return j < 0 ? T(0) : x[j];
}

// wave64 butterfly reductions
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

// block reductions through LDS scratch of >= blockDim.x/64 elements; result broadcast to all
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch)
{
    v = wave_sum(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    T s = 0;
    for (int i = 0; i < nw; ++i) s += scratch[i];
    return s;
}
template <typename T>
__device__ __forceinline__ T block_max(T v, T* scratch)
{
    v = wave_max(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    T s = scratch[0];
    for (int i = 1; i < nw; ++i) s = scratch[i] > s ? scratch[i] : s;
    return s;
}

__device__ __forceinline__ float dsa_log(float v) { return logf(v); }
__device__ __forceinline__ double dsa_log(double v) { return log(v); }
__device__ __forceinline__ float dsa_exp(float v) { return expf(v); }
__device__ __forceinline__ double dsa_exp(double v) { return exp(v); }
__device__ __forceinline__ float dsa_pow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double dsa_pow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float dsa_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double dsa_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float dsa_cos(float v) { return cosf(v); }
__device__ __forceinline__ double dsa_cos(double v) { return cos(v); }
__device__ __forceinline__ float dsa_sin(float v) { return sinf(v); }
__device__ __forceinline__ double dsa_sin(double v) { return sin(v); }
__device__ __forceinline__ float dsa_log10(float v) { return log10f(v); }
__device__ __forceinline__ double dsa_log10(double v) { return log10(v); }

// spectrum formatter of spec.py:123-132 applied to s = |X|^2 + eps (already floored)
template <typename T>
__device__ __forceinline__ T spec_format(T s, int fmt)
{
    switch (fmt) {
    case DSA_SPEC_DB: return T(10) * dsa_log10(s);
    case DSA_SPEC_LOGMAG: return T(0.5) * dsa_log(s);
    case DSA_SPEC_MAG: return dsa_sqrt(s);
    default: return s;
    }
}

}  // namespace dsa
