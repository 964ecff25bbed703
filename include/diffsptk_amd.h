/*
 * diffsptk_amd -- C-ABI of the MI355X (gfx950) device backend for the STFT -> mel-cepstrum
 * and LPC analysis hot path of sp-nitech/diffsptk (v4.0.0).
 *
 * The reference is a pure-Python library over PyTorch: it has NO FFI/plugin interface.  Its
 * operator boundary is the static `_forward(x, **precomputed)` method of every module
 * (diffsptk/modules/base.py:38-101).  Each entry point below is what a binding for one such
 * `_forward` (and its autograd-derived backward) binds; the reference
 * symbol it replaces is cited per function.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HIP, same process / same HIP runtime as the caller)
 *    unless the name ends in `_host`; the caller (PyTorch) allocates all buffers;
 *  - tensors are dense row-major ("contiguous"); leading batch dims are flattened by the caller;
 *  - a ZERO count (B, F, n = 0: an empty batch, which the reference's ATen ops accept) is a successful no-op: the sizes are
 *    validated, the pointers are not looked at (an empty tensor has no storage; its data pointer is NULL);
 *  - `dtype`: DSA_F32 or DSA_F64 (the reference supports both; CI runs float64);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous
 *    on that stream, re-entrant, and keep no mutable global state: the library keeps NO device memory (the only
 *    allocations it makes are transient, stream-ordered hipMallocAsync / hipFreeAsync pairs inside one call: the partial
 *    spans of dsa_stft_bwd / dsa_istft_fwd for frame geometries other than L = 400, P = 80 / 160 -- those run the
 *    one-launch kernel that needs none -- and the zero waveform of the generic inverse path).
 *    The tuned kernels need two kinds of workspace, both owned by the caller:
 *      `scratch`  DSA_SCRATCH_BYTES of device memory per call (work-queue counters of the persistent kernels),
 *                 zeroed by the library on `stream`; it must not be shared by calls that can overlap in time (one
 *                 buffer per stream, or one fresh buffer per call from a stream-ordered allocator).  scratch = NULL
 *                 selects the kernel family that needs none (DSA_ALGO_TUNED then fails with
 *                 DSA_ERR_INVALID_ARGUMENT);
 *      `images`   per-CONFIGURATION constants prepared once (dsa_mcep_prepare), read-only afterwards and
 *                 shareable by any number of concurrent calls on the same device;
 *    the only per-process state is, per kernel, the set of devices whose dynamic-LDS limit has been raised;
 *  - return value: DSA_OK (0) or a negative dsa_status; dsa_last_error() gives the message of
 *    the last failure on the calling thread.  Nothing throws across the boundary.
 *  - argument VALIDATION of user-facing options (ValueError text etc.) is done by the host
 *    layer exactly like the reference's `_check`; the library re-checks only what would make a
 *    launch unsafe.
 */
#ifndef DIFFSPTK_AMD_H
#define DIFFSPTK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSA_VERSION 129 /* 0.2.3: + DSA_ALGO_RESERVE_CUS; a zero count is a no-op before any pointer check; 0.2.2: + dsa_mcep_newton_glogx_h (glogx of the 48 kHz analysis in one pass after the sweep; dsa_mcep_newton_resid_h_bwd takes glogx = NULL); twin workgroups in dsa_mcep_newton_steps; 0.2.1: + dsa_mcep_resid_bwd_images_bytes / _prepare, dsa_mcep_newton_resid_h_bwd (the 48 kHz analysis with a gradient: the step's backward in two launches); wide tiles in dsa_mcep_newton_steps; 0.2.0: + dsa_mcep_newton_steps, dsa_stft_mcep_opts_fwd, DSA_ALGO_OVERLAPPED_LAUNCHES, DSA_ALGO_PAD_MODE, packed STFT kernels for fft_length 1024 / 2048 and for every pad mode at 512; 0.1.9: + dsa_mgcep_step_solve, dsa_mgcep_step_bwd_h, dsa_mcep_resid_images_bytes / _prepare, dsa_mcep_newton_resid_h; 0.1.8: + dsa_frame_window_lpc_bwd, DSA_LPC_EXACT_LAGSUMS, DSA_ALGO_HIST_HAS_RT; 0.1.7: + dsa_gnorm_fwd, dsa_mgcep_gain, DSA_LPC_SCRATCH_IS_CLEAN; 0.1.6: + dsa_mcep_newton_update_bwd; 0.1.5: + dsa_mcep_newton_resid; 0.1.4: dsa_stft_mcep_fwd (STFT -> mel-cepstrum in one launch), dsa_rows_gemm, dsa_rows_ew, dsa_mcep_newton_update */

typedef enum {
    DSA_OK = 0,
    DSA_ERR_INVALID_ARGUMENT = -1,
    DSA_ERR_UNSUPPORTED = -2,
    DSA_ERR_LAUNCH = -3,
    DSA_ERR_NO_DEVICE = -4
} dsa_status;

enum { DSA_F32 = 0, DSA_F64 = 1 };
/* size of the per-call `scratch` the tuned persistent kernels draw their work items through */
#define DSA_SCRATCH_BYTES 64
/* F.pad modes of Frame (frame.py:134-137) */
enum { DSA_PAD_CONSTANT = 0, DSA_PAD_REFLECT = 1, DSA_PAD_REPLICATE = 2, DSA_PAD_CIRCULAR = 3 };
/* fftr.py:110-121 */
enum { DSA_FFTR_COMPLEX = 0, DSA_FFTR_REAL = 1, DSA_FFTR_IMAG = 2, DSA_FFTR_AMPLITUDE = 3, DSA_FFTR_POWER = 4 };
/* spec.py:123-132 (+ complex pass-through of stft.py:211-222) */
enum { DSA_SPEC_DB = 0, DSA_SPEC_LOGMAG = 1, DSA_SPEC_MAG = 2, DSA_SPEC_POWER = 3, DSA_SPEC_COMPLEX = 4,
       /* the inverse real transform's weights c_k / nfft (c = 1 at k = 0 and nfft/2, else 2) folded into the STFT
        * kernels: in dsa_stft_bwd / dsa_istft_fwd the complex spectrogram to INVERT (istft.py:186-193) is weighted
        * while it is loaded; in dsa_stft_fwd the complex output is weighted (the adjoint: the backward of the ISTFT) */
       DSA_SPEC_COMPLEX_INV = 5 };
/* acorr.py:94-107 */
enum { DSA_ACORR_NAIVE = 0, DSA_ACORR_NORMALIZED = 1, DSA_ACORR_BIASED = 2, DSA_ACORR_UNBIASED = 3 };
/* kernel selection: AUTO picks the tuned gfx950 kernel when the configuration allows it */
enum { DSA_ALGO_AUTO = 0, DSA_ALGO_GENERIC = 1, DSA_ALGO_TUNED = 2 };
/* OR-ed into `algo` of dsa_mcep_fwd: the caller guarantees that `scratch` is ZERO on entry (zeroed once, at allocation) and is used
 * by one call at a time; the library then skips its per-call reset (a 5 us fill launch and a kernel boundary per call) -- the
 * persistent kernel leaves the counters zeroed when its last wave retires, with or without this flag. */
#define DSA_ALGO_SCRATCH_IS_CLEAN 0x100
/* OR-ed into `algo` of dsa_mcep_bwd: `scratch` is DSA_MCEP_BWD_WORKSPACE_BYTES long (the counters + a hand-over area).  The tuned
 * backward then cuts a short last round of tiles into pieces of Newton steps that pass their state through it, so that every wave
 * slot ends the launch busy (3 200 tiles on 1 024 slots: 3.3 rounds instead of 4).  Without the flag: DSA_SCRATCH_BYTES, no split. */
#define DSA_ALGO_SCRATCH_HAS_WORKSPACE 0x200
#define DSA_MCEP_BWD_WORKSPACE_BYTES (DSA_SCRATCH_BYTES + 512 * 16 * 32 * 4)
/* OR-ed into `algo` of dsa_mcep_fwd / dsa_stft_mcep_fwd / dsa_mcep_bwd (0.1.8): `mc_hist` continues behind the (n_iter + 1, F, M + 1)
 * iterates with (n_iter, F, 2 M + 1) float32 rows -- every Newton step's rt = e E (mcep.py:212-215).  The tuned forward writes them, the
 * tuned backward reads them instead of recomputing its second forward chain -- and runs as the two-waves-per-SIMD kernel
 * (csrc/mcep_mfma_bwd2_f16.h: 1.56 -> 1.1-1.2 ms per 204 800 frames for 196 more bytes per frame and step).  Both calls of a pair must agree on the flag; the generic kernels ignore the extra room. */
#define DSA_ALGO_HIST_HAS_RT 0x400
/* OR-ed into `algo` of dsa_mcep_fwd / dsa_stft_mcep_fwd (0.2.0): the caller alternates consecutive, independent launches between TWO
 * streams (each with its own `scratch`).  A launch whose tiles do not fill a whole number of rounds of the chip's 2 048 wave slots
 * (204 800 frames = 12 800 tiles = 6.25 rounds) ends in a short round.  Without the flag that round runs one wave per SIMD on every
 * CU (the fastest way to finish ONE launch: 0.8 of a full round's time, the rest of the chip idle).  With it the short round is
 * packed onto the first ceil(rest / 8) workgroups at two waves per SIMD and every other workgroup EXITS when the shared tiles run
 * out, so that the next launch's workgroups -- waiting on the other stream for LDS -- start on the freed CUs (measured: 0.5945 ->
 * 0.5685 ms per 204 800 frames over 200 steps; the ideal is 6.25 rounds per launch instead of 6.8).  Same tiles, same arithmetic,
 * same results bit for bit; a lone launch gets 0.2 of a round slower. */
#define DSA_ALGO_OVERLAPPED_LAUNCHES 0x800
/* OR-ed into `algo` of dsa_stft_mcep_fwd (0.2.0): the padding mode of Frame (frame.py:130-137; DSA_PAD_*, 0 = constant) for the
 * samples a frame reads outside its utterance -- the one-launch step then covers ShortTimeFourierTransform(mode=...) too. */
#define DSA_ALGO_PAD_MODE(m) (((m) & 3) << 12)
/* OR-ed into `algo` of dsa_mcep_fwd / dsa_stft_mcep_fwd / dsa_stft_mcep_opts_fwd (0.2.3): the persistent launch leaves n (<= 63) of the
 * 256 CUs free.  A persistent workgroup fills its CU's LDS and registers, so a kernel of ANOTHER stream that has to run beside the
 * launch -- RCCL's all-gather of the previous batch's features -- could otherwise only start in the launch's tail and the
 * next launch would queue behind it: the exchange would be serial with the analysis instead of hidden behind it.  Same tiles, same
 * arithmetic, same bits.  The launcher leaves at least n and up to 2 n CUs where that costs no further round of tiles.  Measured with a stand-in
 * collective (profiles/r06_reserve_cus_ab.txt): what matters most is WHEN the caller waits for the collective -- two batches later, 8 CUs
 * suffice (0.618 ms per step against 0.584 alone); one batch later the two run in series whatever is reserved below 16.  dist.py sets it. */
#define DSA_ALGO_RESERVE_CUS(n) (((n) & 63) << 16)
#define DSA_ALGO_RESERVED_CUS(algo) (((algo) >> 16) & 63)

int dsa_version(void);
const char* dsa_last_error(void);
/* number of visible HIP devices, or a negative dsa_status */
int dsa_device_count(void);
/* name of the kernel family the last call on this thread dispatched to (for tests/profiles) */
const char* dsa_last_kernel(void);

/* N = (T-1)/P + 1 frames for T >= 1 (frame.py:130-138: padded length is T + L - 1). */
int64_t dsa_num_frames(int64_t T, int32_t P);

/* ------------------------------------------------------------------ a1  Frame
 * Frame._forward, diffsptk/modules/frame.py:120-141.
 * x:(B,T) -> y:(B,N,L); bit-exact gather copy (zmean subtracts the frame mean). */
int dsa_frame_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t center,
                  int32_t zmean, int32_t pad_mode, int32_t dtype, void* y, void* stream);
/* gy:(B,N,L) -> gx:(B,T): deterministic overlap-add (adjoint of pad + unfold + zmean). */
int dsa_frame_bwd(const void* gy, int64_t B, int64_t T, int32_t L, int32_t P, int32_t center,
                  int32_t zmean, int32_t pad_mode, int32_t dtype, void* gx, void* stream);

/* ------------------------------------------------------------------ a2  Window
 * Window._forward, window.py:185-193.  x:(F,L) * w:(L) -> y:(F,L2), zero padded (L2>=L) or
 * cropped (L2<L, F.pad with a negative amount). */
int dsa_window_fwd(const void* x, int64_t F, int32_t L, const void* w, int32_t L2, int32_t dtype,
                   void* y, void* stream);
/* gy:(F,L2) -> gx:(F,L) and, if gw != NULL, gw:(L) += sum_f gy*x (learnable window). */
int dsa_window_bwd(const void* gy, const void* x, int64_t F, int32_t L, const void* w, int32_t L2,
                   int32_t dtype, void* gx, void* gw, void* stream);

/* ------------------------------------------------------------------ a3  fftr
 * RealValuedFastFourierTransform._forward, fftr.py:136-151 (torch.fft.rfft + formatter).
 * x:(F,len_in) zero-padded/cropped to nfft -> y:(F,nfft/2+1) real, or interleaved (re,im)
 * pairs for DSA_FFTR_COMPLEX.  twiddle:(nfft,2) = (cos, -sin)(2 pi m / nfft), device. */
int dsa_fftr_fwd(const void* x, int64_t F, int32_t len_in, int32_t nfft, int32_t out_format,
                 const void* twiddle, int32_t dtype, void* y, void* stream);
int dsa_fftr_bwd(const void* gy, const void* x, int64_t F, int32_t len_in, int32_t nfft,
                 int32_t out_format, const void* twiddle, int32_t dtype, void* gx, void* stream);

/* ------------------------------------------------------------------ a4  Spectrum
 * Spectrum._forward, spec.py:152-178.  b:(F,lb) and/or a:(F,la) (either may be NULL, not both).
 * y:(F,nfft/2+1) = format(max(|K B/A|^2 + eps, floor)). */
int dsa_spec_fwd(const void* b, int32_t lb, const void* a, int32_t la, int64_t F, int32_t nfft,
                 double eps, int32_t use_floor, double relative_floor_db, int32_t out_format,
                 const void* twiddle, int32_t dtype, void* y, void* stream);
int dsa_spec_bwd(const void* gy, const void* b, int32_t lb, const void* a, int32_t la, int64_t F,
                 int32_t nfft, double eps, int32_t use_floor, double relative_floor_db,
                 int32_t out_format, const void* twiddle, int32_t dtype, void* gb, void* ga,
                 void* stream);

/* ------------------------------------------------------------------ a5  STFT (fused a1+a2+a3+a4)
 * ShortTimeFourierTransform._forward, stft.py:237-241 = spec(window(frame(x))).
 * x:(B,T), w:(L) window table -> y:(B,N,nfft/2+1) (x2 interleaved for DSA_SPEC_COMPLEX).
 * One kernel: each waveform sample is read from HBM once, frames overlap in LDS. */
int dsa_stft_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft,
                 const void* w, const void* twiddle, int32_t center, int32_t zmean,
                 int32_t pad_mode, double eps, int32_t use_floor, double relative_floor_db,
                 int32_t out_format, int32_t dtype, int32_t algo, void* y, void* stream);
/* gy like y -> gx:(B,T); if gw != NULL also gw:(L) (learnable window, stft.py:73-76). */
int dsa_stft_bwd(const void* gy, const void* x, int64_t B, int64_t T, int32_t L, int32_t P,
                 int32_t nfft, const void* w, const void* twiddle, int32_t center, int32_t zmean,
                 int32_t pad_mode, double eps, int32_t use_floor, double relative_floor_db,
                 int32_t out_format, int32_t dtype, int32_t algo, void* gx, void* gw, void* stream);

/* ------------------------------------------------------------------ a6/a7  frequency transform
 * FrequencyTransform._forward freqt.py:141-143 and CoefficientsFrequencyTransform._forward
 * mcep.py:286-288: out:(F,L2) = c:(F,L1) @ A:(L1,L2).  (bwd: gc = gout @ A^T) */
int dsa_freqt_fwd(const void* c, int64_t F, int32_t L1, const void* A, int32_t L2, int32_t dtype,
                  void* out, void* stream);
int dsa_freqt_bwd(const void* gout, int64_t F, int32_t L1, const void* A, int32_t L2,
                  int32_t dtype, void* gc, void* stream);

/* ------------------------------------------------------------------ f1  mel filter bank
 * MelFilterBankAnalysis._forward, fbank.py:306-321.  x:(F,K) power spectrum, H:(K,C) filter weights:
 *   y:(F,C) = glog(max(s @ H, floor)), s = x (use_power) or sqrt(x), glog = log (gamma 0) or (y^gamma-1)/gamma;
 *   E:(F) = log((2 sum_{0<k<K-1} x_k + x_0 + x_{K-1}) / (2 (K-1)))   (E may be NULL).
 * MFCC (mfcc.py:244-256) = this + dsa_freqt_fwd with the DCT-II matrix times the liftering vector. */
int dsa_fbank_fwd(const void* x, int64_t F, int32_t K, const void* H, int32_t C, double floor, double gamma,
                  int32_t use_power, int32_t dtype, void* y, void* E, void* stream);
int dsa_fbank_bwd(const void* gy, const void* gE, const void* x, int64_t F, int32_t K, const void* H, int32_t C,
                  double floor, double gamma, int32_t use_power, int32_t dtype, void* gx, void* stream);
/* MelFrequencyCepstralCoefficientsAnalysis._forward mfcc.py:244-256 in one launch: z:(F,Mo) = glog(max(s H, floor)) W,
 * W:(C,Mo) = DCT-II x truncation x liftering vector (device); E:(F) log energy as dsa_fbank_fwd (may be NULL).  The
 * filter-bank outputs are not materialised; the backward is dsa_freqt_bwd (through W) followed by dsa_fbank_bwd. */
int dsa_fbank_dct_fwd(const void* x, int64_t F, int32_t K, const void* H, int32_t C, const void* W, int32_t Mo, double floor,
                      double gamma, int32_t use_power, int32_t dtype, void* z, void* E, void* stream);
/* The two stages above in ONE launch -- STFT (stft.py:148-152, power format, constant padding) and
 * MelFilterBankAnalysis._forward (fbank.py:306-321, out_format "y") without the (B, N, 257) spectrum ever reaching
 * memory: x:(B,T) -> y:(B, N, C) = glog(max(s @ H, floor)), s = |STFT|^2 + eps (use_power) or its square root.
 * H enters as the per-lane `plan` (device, DSA_FBANK_PLAN_FLOATS float32) that dsa_fbank_scan_plan builds ON THE
 * HOST from H:(257, C) float64 row-major: it exists when every bin feeds at most two ADJACENT channels, in
 * ascending order along the bins (the triangular mel / auditory filters of fbank.py:232-291; C <= 126) -- otherwise
 * DSA_ERR_UNSUPPORTED, and the two-call path above serves the matrix.  The fused kernel covers float32,
 * fft_length 512, frame_length 400, even frame_period (else DSA_ERR_UNSUPPORTED).
 * Its BACKWARD needs no spectrum either: dsa_fbank_bins_bwd turns the cotangent gy:(F, C) of the output and the saved
 * output y:(F, C) into the cotangent g:(F, K) of the filter bank's input -- g[k] = w0 Q(c_k) + w1 Q(c_k + 1),
 * Q(c) = gy_c * dglog/ds at s = glog^-1(y_c), 0 where the floor clamped (fbank.py:312-321 differentiated) -- through the
 * per-bin `table` (device, 4 K float32: {bits of c_k, w0, w1, 0} per bin) that dsa_fbank_bins_plan builds ON THE HOST from
 * H:(K, C) float64 (DSA_ERR_UNSUPPORTED when a bin feeds more than two adjacent channels); g then enters dsa_stft_bwd as
 * the cotangent of the power (use_power) or magnitude spectrum.  float32. */
#define DSA_FBANK_PLAN_FLOATS 2048
int dsa_fbank_scan_plan(const double* H_host, int32_t K, int32_t C, float* plan_host);
int dsa_stft_fbank_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* w,
                       const void* twiddle, int32_t center, double eps, const void* plan, int32_t C, double floor,
                       double gamma, int32_t use_power, int32_t dtype, void* y, void* stream);
int dsa_fbank_bins_plan(const double* H_host, int32_t K, int32_t C, float* table_host);
int dsa_fbank_bins_bwd(const void* gy, const void* y, int64_t F, int32_t K, int32_t C, const void* table, double floor,
                       double gamma, int32_t dtype, void* g, void* stream);

/* ------------------------------------------------------------------ f2  inverse path
 * RealValuedInverseFastFourierTransform ifftr.py:131-142, Unframe unframe.py:164-211, InverseShortTimeFourier-
 * Transform istft.py:186-193.  irfft is the ADJOINT of rfft applied to G_k = c_k / N Y_k (c = 1 at DC and
 * Nyquist, else 2), and overlap-add is the adjoint of framing: the inverse path runs on the backward entries
 * above (dsa_fftr_bwd / dsa_frame_bwd / dsa_stft_bwd with complex cotangents) plus these two helpers.
 *   dsa_irfft_scale: y:(F,nfft/2+1) complex pairs -> out = c_k / nfft * y
 *   dsa_div_rows:    out:(B,T) = x / (d:(T) + eps)   (d = overlap-added squared window, eps = 1e-16) */
int dsa_irfft_scale(const void* y, int64_t F, int32_t nfft, int32_t dtype, void* out, void* stream);
int dsa_div_rows(const void* x, int64_t B, int64_t T, const void* d, double eps, int32_t dtype, void* out, void* stream);
/* InverseShortTimeFourierTransform._forward istft.py:186-193 in one call: y:(B,N,nfft/2+1) complex pairs, N =
 * dsa_num_frames(T, P) -> out:(B,T) = overlap-add(w * irfft(y)[:L]) / (d + d_eps); w:(L) synthesis window, d:(T) the
 * overlap-added squared window (unframe.py:203-205), twiddle as dsa_stft_fwd.  Float32 / fft_length 512 runs on the
 * tuned STFT backward kernels (inverse weights while loading; the division happens as the samples are stored). */
int dsa_istft_fwd(const void* y, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* w,
                  const void* twiddle, int32_t center, const void* d, double d_eps, int32_t dtype, int32_t algo,
                  void* out, void* stream);
/* GriffinLim._forward griffin.py:263-284: the element-wise part of one phase-reconstruction step between
 * z -> dsa_stft_bwd (inverse) -> dsa_stft_fwd (complex) -> t.  y:(B,N,K) power spectrogram; t:(B,Nt,K) complex
 * pairs (Nt >= N, surplus frames dropped; NULL = initial step with phase:(B,N,K) or NULL for zeros);
 * t_prev, d_prev:(B,N,K) complex pairs, updated in place; first != 0 on the first step; z:(B,N,K) complex pairs
 * = sqrt(y + 1e-16) c / (|c| + eps), the next spectrogram to invert. */
int dsa_griffin_update(const void* t, int64_t B, int64_t Nt, int64_t N, int32_t K, const void* y, const void* phase,
                       void* t_prev, void* d_prev, int32_t first, double alpha, double beta, double gamma, double eps,
                       int32_t dtype, void* z, void* stream);

/* ------------------------------------------------------------------ f3  cepstral analysis
 * CepstralAnalysis._forward fftcep.py:116-136 (improved cepstral method).  x:(F, L/2+1) power spectra ->
 * out:(F, M+1).  A:(L/2+1, L/2+1) = c_k cos(2 pi k n / L) (c = 1 at k = 0 and L/2, else 2), device, row-major:
 * every transform of the reference (irfft / hfft / ihfft of even real sequences) is a product with it.
 * masks:(F, n_iter, ceil((L/2+1)/64)) uint64 bit sets of the clamp pattern, written by the forward for the
 * backward (may be NULL when n_iter == 0 or no gradient is needed). */
int dsa_fftcep_fwd(const void* x, int64_t F, int32_t fft_length, int32_t cep_order, const void* A, double accel,
                   int32_t n_iter, int32_t dtype, void* out, void* masks, void* stream);
int dsa_fftcep_bwd(const void* gout, const void* x, int64_t F, int32_t fft_length, int32_t cep_order, const void* A,
                   double accel, int32_t n_iter, const void* masks, int32_t dtype, void* gx, void* stream);

/* ------------------------------------------------------------------ a8-a10  mel-cepstral analysis
 * MelCepstralAnalysis._forward, mcep.py:189-224 (incl. symmetric_toeplitz / hankel,
 * utils/private.py:291-302, and the torch.linalg.solve call at mcep.py:221).
 * X:(F,nfft/2+1) power spectrum -> mc:(F,M+1).  The host composes the reference's linear maps
 * once per configuration (float64, then cast):
 *   G:(H+1,M+1)  = irfft-with-halved-ends  o freqt          (mcep.py:204-207)
 *   D:(M+1,H+1)  = ifreqt o Re(rfft(., nfft))               (mcep.py:210-211)
 *   E:(H+1,2M+1) = irfft o rfreqt                           (mcep.py:214-215)
 *   alpha_vec:(M+1) = (-alpha)^i                            (mcep.py:179-181)
 * so that one Newton step is  d = mc D ; e = exp(log X - 2 d) ; rt = e E ;
 * mc += solve(T(rt[:M+1]) + H(rt), rt[:M+1] - alpha_vec).
 * mc_hist: NULL or (n_iter+1, F, M+1) receiving mc after 0..n_iter steps (saved for backward).
 * images / scratch: see Conventions.  The tuned gfx950 kernel (float32, fft_length 512, cep_order 24) consumes G, D, E
 * as binary16 hi/lo operand images in matrix-core lane order: dsa_mcep_images_bytes() is their size (0 when the
 * configuration has no tuned kernel: images may then be NULL) and dsa_mcep_prepare() writes them -- once per
 * configuration, what MelCepstralAnalysis._precompute (mcep.py:133-187) is to the reference.  With
 * DSA_ALGO_AUTO the tuned kernel runs iff the configuration allows it AND images and scratch are given. */
int64_t dsa_mcep_images_bytes(int32_t nfft, int32_t M, int32_t dtype);
int dsa_mcep_prepare(const void* G, const void* D, const void* E, int32_t nfft, int32_t M, int32_t dtype,
                     void* images, void* stream);
int dsa_mcep_fwd(const void* X, int64_t F, int32_t nfft, int32_t M, int32_t n_iter, const void* G,
                 const void* D, const void* E, const void* alpha_vec, int32_t dtype, int32_t algo,
                 const void* images, void* scratch, void* mc, void* mc_hist, void* stream);
/* Gain normalisation of generalized cepstra (gnorm.py:102-112) and its inverse (ignorm.py:99-109), forward, one launch each:
 *   inverse = 0:  x:(F,n) -> (K, x1 / (1 + gamma x0)),  K = (1 + gamma x0)^(1/gamma)   (gamma = 0: (exp x0, x1))
 *   inverse = 1:  y:(F,n) -> ((K^gamma - 1) / gamma, y1 K^gamma)                          (gamma = 0: (log K, y1)) */
int dsa_gnorm_fwd(const void* x, int64_t F, int32_t n, double gamma, int32_t inverse, int32_t dtype, void* out, void* stream);
/* The gain of a Newton step of the mel-generalized analysis joined to its coefficients (mgcep.py:213-215, 221, 231-233):
 * b:(F, M+1) = (sqrt(r_0 + gamma sum_m r_{m+1} b_eps_m), b_join),  r:(F, M+1), b_eps, b_join:(F, M) (the same rows at gamma = -1,
 * the coefficients before / after the step's update otherwise). */
int dsa_mgcep_gain(const void* r, const void* b_eps, const void* b_join, int64_t F, int32_t M, double gamma, int32_t dtype, void* b,
                   void* stream);
/* The solve-and-update of a Newton step of MelCepstralAnalysis (mcep.py:216-222) at a geometry without a tuned kernel:
 * mc_out:(F,n) = mc_in + solve(T(rt[:, :n]) + H(rt), rt[:, :n] - alpha_vec), rt:(F, 2n-1), n = cep_order + 1 in [2, 55], float32.
 * 16 systems per wave on the float32 matrix instruction, no pivoting; a system that meets a non-positive pivot is re-solved with
 * row pivoting by a second launch (csrc/thsolve_quad.hip).  mc_out may be mc_in. */
int dsa_mcep_newton_update(const void* rt, int64_t F, int32_t n, const void* alpha_vec, int32_t dtype, const void* mc_in,
                           void* mc_out, void* stream);
/* Its backward.  mc_in = NULL in the call above leaves the solution s alone in mc_out; with gs:(F,n) the cotangent of s this writes
 * grt:(F, 2n-1), the cotangent of rt: u = A^-1 gs on the same batched solve (A is symmetric; u:(F,n) is caller workspace), then
 * grt[k] = -sum_{i+j=k} u_i s_j - [k<n] sum_{|i-j|=k} u_i s_j + [k<n] u_k   (Hankel part, Toeplitz part, right-hand side). */
int dsa_mcep_newton_update_bwd(const void* gs, const void* rt, const void* sol, int64_t F, int32_t n, int32_t dtype, void* u,
                               void* grt, void* stream);
/* The spectral half of the same Newton step (mcep.py:210-215) in one launch:
 *   rt:(F, 2n-1) = exp(logx - 2 mc D) E,   logx:(F,K) = log X, mc:(F,n), D:(n x K, row stride ldd), E:(K x (2n-1), row stride lde),
 * n = cep_order + 1 in [3, 55], K >= 4, float32.  e = exp(.) is produced chunk by chunk in the operand layout of the second
 * product (v_mfma_f32_16x16x4_f32 for both) and never reaches memory (csrc/rows_gemm.hip:mcep_resid_mfma_kernel). */
int dsa_mcep_newton_resid(const void* logx, int64_t F, int32_t K, const void* mc, int32_t n, const void* D, int32_t ldd,
                          const void* E, int32_t lde, int32_t dtype, void* rt, void* stream);
/* (0.1.9) The same launch with both products as 3-term binary16 splits on the matrix pipe (csrc/mcep_resid_f16.h: 12 + 21 binary16
 * products per 32 bins and 16 frames at order 49 instead of 82 float32 ones): `images` = dsa_mcep_resid_images_bytes(K, n) bytes of
 * caller-owned device memory filled ONCE per configuration by dsa_mcep_resid_prepare from D (n x K) and E (K x 2n - 1); logx holds
 * natural logarithms as for dsa_mcep_newton_resid. */
int64_t dsa_mcep_resid_images_bytes(int32_t K, int32_t n);
int dsa_mcep_resid_prepare(const void* D, int32_t ldd, const void* E, int32_t lde, int32_t K, int32_t n, int32_t dtype, void* images,
                           void* stream);
/* (0.2.0) ALL n_iter Newton steps of mcep.py:208-222 in ONE persistent launch (csrc/mcep_big_f16.h): per step the two products of
 * dsa_mcep_newton_resid_h (same images, bit-identical rt) and the solve-and-update of dsa_mcep_newton_update, rt and mc staying on
 * chip; mc_in:(F, n) the start (mc0 = logx G), mc_out:(F, n) (may be mc_in).  Orders n - 1 in 32 .. 54 (the 48 kHz set-ups fft_length
 * 2048 / order 49 and 1024 / order 34 among them); DSA_ERR_UNSUPPORTED otherwise -- alternate dsa_mcep_newton_resid_h and dsa_mcep_newton_update then. */
int dsa_mcep_newton_steps(const void* logx, int64_t F, int32_t K, const void* mc_in, int32_t n, const void* images, const void* alpha_vec,
                          int32_t n_iter, int32_t dtype, void* mc_out, void* stream);
int dsa_mcep_newton_resid_h(const void* logx, int64_t F, int32_t K, const void* mc, int32_t n, const void* images, int32_t dtype,
                            void* rt, void* stream);
/* (0.2.1) The ADJOINT of dsa_mcep_newton_resid_h -- what autograd derives from mcep.py:210-215 -- in one launch on the binary16 matrix
 * pipe (csrc/mcep_resid_bwd_f16.h), orders n - 1 in 32 .. 54 (dsa_mcep_resid_bwd_images_bytes returns 0 otherwise: keep the composed
 * gradient there): given grt:(F, 2n - 1), the cotangent of rt = exp(logx - 2 mc D) E,
 *   glogx:(F, K) += ebar * e,   gmc:(F, n) = -2 (ebar * e) D^T,   ebar = grt E^T,  e = exp(logx - 2 mc D) recomputed from the iterate
 * (glogx is read-modify-written: the caller zeroes it before the first step of the reverse sweep; gmc is overwritten).  `images` =
 * dsa_mcep_resid_bwd_images_bytes(K, n) bytes of caller-owned device memory filled ONCE per configuration by
 * dsa_mcep_resid_bwd_prepare from D (n x K) and E (K x 2n - 1).  With dsa_mcep_newton_update_bwd (the solve on the cotangent and
 * the diagonal sums: grt) this is a Newton step's whole backward: two launches, no intermediate of the forward kept but the iterates,
 * the rt rows and the solutions. */
int64_t dsa_mcep_resid_bwd_images_bytes(int32_t K, int32_t n);
int dsa_mcep_resid_bwd_prepare(const void* D, int32_t ldd, const void* E, int32_t lde, int32_t K, int32_t n, int32_t dtype, void* images,
                               void* stream);
int dsa_mcep_newton_resid_h_bwd(const void* logx, int64_t F, int32_t K, const void* mc, int32_t n, const void* grt, const void* images,
                                int32_t dtype, void* glogx, void* gmc, void* stream);
/* (0.2.2) The sum over the Newton steps that dsa_mcep_newton_resid_h_bwd accumulates into glogx, formed in ONE pass over the bins after
 * the reverse sweep instead (autograd of mcep.py:210-215 summed over mcep.py:208-222's iterations):
 *   glogx[f, k] = sum_s (sum_j grts[s][f][j] E[k][j]) * exp(logx[f][k] - 2 sum_c mcs[s][f][c] D[c][k])
 * mcs:(n_iter, F, n) the iterates the steps started from, grts:(n_iter, F, 2n - 1) the cotangents of their rt rows
 * (dsa_mcep_newton_update_bwd's `grt`, kept per step), images of dsa_mcep_resid_bwd_prepare; glogx:(F, K) is WRITTEN.  The sweep's
 * launches then pass glogx = NULL to dsa_mcep_newton_resid_h_bwd (a step moves the (F, K) array once instead of three times).  The
 * steps are summed in the sweep's order on the same values: bit-identical to the in-place accumulation.  DSA_ERR_UNSUPPORTED
 * (no error text): orders outside 32 .. 54, or n_iter above what one workgroup's LDS holds (12 at orders >= 48, 15 below) -- the
 * caller keeps the in-place accumulation. */
int dsa_mcep_newton_glogx_h(const void* logx, int64_t F, int32_t K, const void* mcs, int32_t n, const void* grts, int32_t n_iter,
                            const void* images, int32_t dtype, void* glogx, void* stream);
/* General float32 row product on the matrix instruction, for shapes the kernels behind dsa_freqt_fwd / _bwd do not cover (rows
 * of 512 values and more: the 1025-bin products of the 48 kHz set-ups of utils/public.py:22-104) and for the Newton step of
 * MelCepstralAnalysis (mcep.py:203-215) at geometries without a tuned kernel:
 *   out:(F,N) = op_out( op_in(c):(F,K) x B:(K,N) ),  B = A (K x N, row stride lda) or, with DSA_ROWS_TRANS, A^T (A: N x K, stride lda)
 *   DSA_ROWS_PRO_LOG    op_in = log          (mcep.py:203 feeding :204-207)
 *   DSA_ROWS_EPI_EXPSUB op_out(v) = exp(aux - 2 v), aux:(F,N) with row stride ldaux   (mcep.py:210-212)
 * Exact float32 products, float32 accumulation (v_mfma_f32_16x16x4_f32).  ldo: row stride of out. */
#define DSA_ROWS_PRO_LOG 1
#define DSA_ROWS_EPI_EXPSUB 2
#define DSA_ROWS_TRANS 4
int dsa_rows_gemm(const void* c, int64_t F, int32_t K, const void* A, int32_t lda, int32_t N, int32_t flags, const void* aux,
                  int32_t ldaux, int32_t dtype, void* out, int32_t ldo, void* stream);
/* element-wise companions of the same analysis when a gradient is wanted (the fused forms keep no operands):
 *   op 0: o0 = log(a)            backward (backward = 1): o0 = gy / a
 *   op 1: o0 = exp(a - 2 b)      backward: a = the saved OUTPUT y, o0 = gy y (cotangent of a), o1 = -2 gy y (cotangent of b)   */
int dsa_rows_ew(int32_t op, int32_t backward, const void* a, const void* b, const void* gy, int64_t n, int32_t dtype, void* o0,
                void* o1, void* stream);
/* ShortTimeFourierTransform._forward (stft.py:237-241: frame, window, rfft, |.|^2 + eps) feeding
 * MelCepstralAnalysis._forward (mcep.py:189-224) in ONE launch: x:(B,T) -> mc:(B N, M+1), N = (T-1)/P + 1, without the
 * (B, N, nfft/2+1) power spectrogram's round trip through memory (320 + 100 bytes per frame instead of 1348 + 1128).  The
 * persistent mel-cepstral wave computes the 16 spectra of its tile itself, with the instructions of the packed STFT kernel
 * (bit-identical power values), and takes log2 straight into the registers the Newton iteration keeps them in.
 * Covers float32, frame_length 400, fft_length 512, cep_order 24, power format, constant padding, no zmean / relative floor
 * (else DSA_ERR_UNSUPPORTED: call dsa_stft_fwd and dsa_mcep_fwd).  window:(L), twiddle:(nfft,2), G/D/E/alpha_vec/images/
 * scratch/algo flags as dsa_mcep_fwd; mc_hist: NULL or (n_iter+1, B N, M+1); X_out: NULL or (B N, nfft/2+1) receiving the
 * power spectrogram as a side product (what dsa_mcep_bwd needs when a gradient is wanted). */
/* (0.2.0) The same launch with the options of ShortTimeFourierTransform it covers beyond the plain configuration (stft.py:86-104):
 * zmean (frame.py:139-140), pad_mode (DSA_PAD_*, frame.py:130-137), the relative floor in dB (spec.py:174-176); power format only.
 * Options run own instantiations of the kernel: the plain configuration's code is untouched. */
int dsa_stft_mcep_opts_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* window,
                           const void* twiddle, int32_t center, int32_t zmean, int32_t pad_mode, double eps, int32_t use_floor,
                           double relative_floor_db, int32_t M, int32_t n_iter, const void* G, const void* D, const void* E,
                           const void* alpha_vec, int32_t dtype, int32_t algo, const void* images, void* scratch, void* mc,
                           void* mc_hist, void* X_out, void* stream);
int dsa_stft_mcep_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* window,
                      const void* twiddle, int32_t center, double eps, int32_t M, int32_t n_iter, const void* G,
                      const void* D, const void* E, const void* alpha_vec, int32_t dtype, int32_t algo,
                      const void* images, void* scratch, void* mc, void* mc_hist, void* X_out, void* stream);
/* gradient of the UNROLLED n_iter-step iteration (what autograd gives the reference).
 * gmc:(F,M+1), X, mc_hist as saved by the forward -> gX:(F,nfft/2+1). */
int dsa_mcep_bwd(const void* gmc, const void* X, const void* mc_hist, int64_t F, int32_t nfft,
                 int32_t M, int32_t n_iter, const void* G, const void* D, const void* E,
                 const void* alpha_vec, int32_t dtype, int32_t algo, const void* images, void* scratch,
                 void* gX, void* stream);

/* ------------------------------------------------------------------ f3  mel-generalized cepstral analysis
 * The Toeplitz-plus-Hankel solve of MelGeneralizedCepstralAnalysis.forward, mgcep.py:226-229 (symmetric_toeplitz /
 * hankel of utils/private.py:291-302, torch.linalg.solve): g:(F,n) = solve(T(p) + H(q), r), p:(F,n) the first
 * column of the Toeplitz part, q:(F,2n-1) the anti-diagonals of the Hankel part, r:(F,n); n <= 64, row-pivoted
 * (float32, n = 24: 16 systems per wave without pivoting, as the mel-cepstral kernels solve theirs -- the matrix is positive
 * definite for the analysis' gamma in [-1, 0]; a system whose elimination meets a non-positive pivot is re-solved with row
 * pivoting by a second launch, so arbitrary symmetric systems get the pivoted answer the reference's LAPACK call gives).
 * Backward: cotangent gg:(F,n) -> gp, gq, gr.  The other stages of the analysis are row products against matrices
 * the host composes (dsa_freqt_fwd) around pointwise spectrum arithmetic (modules/mgcep.py). */
int dsa_thsolve_fwd(const void* p, const void* q, const void* r, int64_t F, int32_t n, int32_t dtype, void* g,
                    void* stream);
/* The pointwise spectrum arithmetic of one Newton step, mgcep.py:199-209 (gamma in [-1, 0), not 0), in one pass:
 * x:(F,K) power spectra, b1:(F,M) the current coefficients b[1:], Cr, Ci:((M+1),K) the composed cfreqt -> rfft
 * matrices (tables.mgcep_matrices; row 0 is not read: b[0] = 0 there) -> out:(5,F,K) = pp, qq (X^2 - Y^2), qq 2XY,
 * pp X, pp Y -- the inputs of the row products against Pr, Qr, Qi, Rr, Ri (dsa_freqt_fwd).  Forward only: with a
 * gradient needed the module composes the same arithmetic from differentiable operators.  M <= 64. */
int dsa_mgcep_spectra(const void* x, const void* b1, int64_t F, int32_t fft_length, int32_t M, const void* Cr,
                      const void* Ci, double gamma, int32_t dtype, void* out, void* stream);
int dsa_thsolve_bwd(const void* gg, const void* p, const void* q, const void* g, int64_t F, int32_t n, int32_t dtype,
                    void* gp, void* gq, void* gr, void* stream);
/* The Newton update of mgcep.py:226-230 in one call: b_out = b_in + solve(symmetric_toeplitz(p) + hankel(q), r) with the
 * right-hand side read in place from a wider vector: row f of r starts at r + f * r_stride + r_offset (the step's (F, n + 1)
 * vector with r_offset = 1).  Order 24 in float32 (DSA_ERR_UNSUPPORTED otherwise); b_in and b_out must not alias. */
int dsa_thsolve_update_fwd(const void* p, const void* q, const void* r, int64_t r_stride, int64_t r_offset, int64_t F,
                           int32_t n, int32_t dtype, const void* b_in, void* b_out, void* stream);
/* The same step's spectrum arithmetic AND its five row products in one launch (mgcep.py:199-220; float32, fft_length 512,
 * cep_order <= 24, gamma in [-1, 0)): x:(F,257), b1:(F,M) -> pt:(F,M) = pp Pr[:, :M], qt:(F,2M-1) = (1 + gamma)(.. Qr[:, 2:] + .. Qi[:, 2:]),
 * r:(F,M+1) = pp X Rr + pp Y Ri -- the operands of dsa_thsolve_fwd.  The five spectra never exist in memory (dsa_mgcep_spectra writes
 * them for the float64 / other-size path).  `images`: 17 x 3840 float32, the matrices in matrix-instruction lane order, built by the
 * caller once per configuration (layout: csrc/mgc.hip, mgcep_step_kernel; diffsptk_amd.utils.tables.mgcep_step_images).  Forward only. */
int dsa_mgcep_step(const void* x, const void* b1, int64_t F, int32_t fft_length, int32_t M, double gamma, const void* images,
                   int32_t dtype, void* pt, void* qt, void* r, void* stream);
/* (0.1.9) dsa_mgcep_step_bwd on the binary16 matrix pipe (cep_order 24 / fft_length 512 / float32; csrc/mgcep_step_f16.h): same
 * arguments, `images_bwd_h`: 9 x 22528 binary16 (diffsptk_amd.utils.tables.mgcep_step_bwd_h_images).  66 binary16 products per 32 bins and
 * 16 frames instead of 144 float32 ones. */
int dsa_mgcep_step_bwd_h(const void* x, const void* b1, const void* gpt, const void* gqt, const void* gr, int64_t F,
                         int32_t fft_length, int32_t M, double gamma, const void* images_bwd_h, int32_t dtype, const void* gx_in,
                         void* gx, void* gb1, void* stream);
/* (0.1.9) The WHOLE Newton step in one launch (mgcep.py:199-230; float32, fft_length 512, cep_order 24, gamma in (-1, 0)): the spectrum
 * arithmetic, the five row products as 3-term binary16 splits on the matrix pipe, the 24 x 24 Toeplitz-plus-Hankel solve (block
 * elimination, pivoted re-solve for systems that are not positive definite) and the update: x:(F,257), b1:(F,24) ->
 * b1_out:(F,24) = b1 + solve(toeplitz(pt) + hankel(qt), r[1:]) (b1_out may be b1), r:(F,25) (what the gain of mgcep.py:221 reads).
 * `images_h`: 9 x 16384 binary16 followed by 240 float32 (the matrices at the Nyquist bin), built by the caller once per configuration
 * (diffsptk_amd.utils.tables.mgcep_step_h_buffer; layout: csrc/mgcep_step_f16.h).  `pt`:(F,24), `qt`:(F,47): NULL, or where the system's two generators are kept for a graph (the backward is
 * dsa_thsolve_bwd on (pt, qt, b1_out - b1) followed by dsa_mgcep_step_bwd).  Replaces dsa_mgcep_step + dsa_thsolve_update_fwd (95 + 32 us
 * per 51 200 frames: 80).  `n_steps` >= 1 Newton steps run in the ONE launch (a frame's iteration depends on the frame alone: the
 * coefficients stay in LDS between the steps); r, pt, qt are the LAST step's, `b1_prev`:(F,24) (or NULL) the last step's input. */
int dsa_mgcep_step_solve(const void* x, const void* b1, int64_t F, int32_t fft_length, int32_t M, double gamma, const void* images_h,
                         int32_t dtype, void* b1_out, void* r, void* pt, void* qt, int32_t n_steps, void* b1_prev, void* stream);
/* Backward of dsa_mgcep_step in one launch: cotangents gpt:(F,M), gqt:(F,2M-1), gr:(F,M+1) -> gx:(F,L/2+1) (+ gx_in when not
 * NULL: the spectrum enters every Newton step, so the steps' contributions accumulate; gx_in may be gx) and gb1:(F,M).
 * `images_bwd`: 17 x 4608 float32 built by the caller (diffsptk_amd/utils/tables.py:mgcep_step_bwd_images). */
int dsa_mgcep_step_bwd(const void* x, const void* b1, const void* gpt, const void* gqt, const void* gr, int64_t F,
                       int32_t fft_length, int32_t M, double gamma, const void* images_bwd, int32_t dtype, const void* gx_in,
                       void* gx, void* gb1, void* stream);
/* GeneralizedCepstrumToGeneralizedCepstrum._forward, mgc2mgc.py:333-361 (the FFT formulation of the generalized cepstral
 * transformation), in ONE launch: c1:(F,n_in) gain-normalised generalized cepstra of in_gamma -> c2:(F,out_order+1) of
 * out_gamma through fft(c01, n_fft) -> (1 + g1 C)^(1/g1) -> (|s|^g2 cos(g2 angle s) - 1) / g2 -> ifft(.).real, n_fft a power of
 * two (the row lives in LDS: 8 n_fft bytes in float32, 16 n_fft in float64, <= 150 KB).  `twiddle`: (n_fft, 2) as for dsa_fftr_fwd.
 * `flags` folds the per-row scalar steps MelGeneralizedCepstrumToMelGeneralizedCepstrum wraps around it (mgc2mgc.py:217-300):
 * 1 = gain normalisation with in_gamma before (gnorm.py:99-109), 2 = inverse gain normalisation with out_gamma after
 * (ignorm.py:99-109), 4 = c[1:] *= out_gamma (GammaMultiplication), 8 = c[0] = c[0] out_gamma + 1 (ZerothGammaMultiplication).
 * Forward only: with a gradient needed the module composes dsa_fftr_fwd / element-wise operators / the inverse transform.
 * This is what mgc2mgc (gamma conversion), mgc2sp and the MLSA filter's impulse responses (mglsadf.py:389-527) run on. */
int dsa_gc2gc_fwd(const void* c1, int64_t F, int32_t n_in, int32_t out_order, double in_gamma, double out_gamma,
                  int32_t nfft, const void* twiddle, int32_t flags, int32_t dtype, void* c2, void* stream);
/* Backward of dsa_gc2gc_fwd with flags = 0, one launch: the row c1:(F, n_in) and the cotangent g2:(F, out_order + 1) of c2 ->
 * gc1:(F, n_in).  n_fft a power of two; float32 up to 8192 points, float64 up to 4096 (the three half-length transforms and the
 * half spectrum live in LDS). */
int dsa_gc2gc_bwd(const void* c1, const void* g2, int64_t F, int32_t n_in, int32_t out_order, double in_gamma, double out_gamma,
                  int32_t nfft, const void* twiddle, int32_t dtype, void* gc1, void* stream);

/* ------------------------------------------------------------------ f4  time-variant all-zero filter
 * AllZeroDigitalFilter._forward_efficient, zerodf.py:207-243: the FIR core of the multi-stage / single-stage MLSA filter
 * (mglsadf.py:254-527).  x:(B,T), b:(B,N,M+1) with T = N P -> y:(B,T),
 *   y[t] = sum_k h_t[k] x[t - k + zeroth_index],  h_t = lerp(b[t / P], b[min(t / P + 1, N - 1)], (t % P) / P);
 * ignore_gain divides by the interpolated gain tap (b[.][0], or b[.][M] when zeroth_index = M).
 * Backward: gy, and the forward's x, b, y -> gx:(B,T) and / or gb:(B,N,M+1) (either may be NULL). */
int dsa_zerodf_fwd(const void* x, const void* b, int64_t B, int64_t T, int32_t M, int32_t P, int32_t zeroth_index,
                   int32_t ignore_gain, int32_t dtype, void* y, void* stream);
int dsa_zerodf_bwd(const void* gy, const void* x, const void* b, const void* y, int64_t B, int64_t T, int32_t M, int32_t P,
                   int32_t zeroth_index, int32_t ignore_gain, int32_t dtype, void* gx, void* gb, void* stream);
/* One Taylor stage of the multi-stage MLSA filter (mglsadf.py:356-365: x <- F x / i, y <- y + x) in one launch:
 *   y = scale * zerodf(x; b)   (y may be NULL when only the sum is wanted: the last stage),
 *   ysum = acc + y             (acc and ysum both NULL or both given; they may be the same buffer).
 * Rounded exactly like the three separate operations.  Needs P % 4 == 0 and M >= 16 (DSA_ERR_UNSUPPORTED otherwise:
 * use dsa_zerodf_fwd and element-wise launches). */
int dsa_zerodf_taylor_fwd(const void* x, const void* b, int64_t B, int64_t T, int32_t M, int32_t P, int32_t zeroth_index,
                          double scale, const void* acc, int32_t dtype, void* y, void* ysum, void* stream);
/* Backward of one Taylor stage (x_i = F x_{i-1} / i, y = sum x_i): with G the cotangent that reaches x_i, one call gives
 *   G_out = gy + scale * F^T G      (the cotangent that reaches x_{i-1}; gy: the cotangent of y; G_out must not alias G),
 *   gb   += scale * dF(x_{i-1})^T G (accumulated over the stages: the caller zeroes gb once; may be NULL).
 * Same shape limits as dsa_zerodf_taylor_fwd. */
int dsa_zerodf_taylor_bwd(const void* G, const void* x, const void* b, int64_t B, int64_t T, int32_t M, int32_t P,
                          int32_t zeroth_index, double scale, const void* gy, int32_t dtype, void* G_out, void* gb, void* stream);

/* ------------------------------------------------------------------ a11  autocorrelation
 * Autocorrelation._forward, acorr.py:110-120.  x:(F,L) -> r:(F,M+1).  Computed as direct lag
 * sums (the reference's irfft(|rfft(x, L+M)|^2) is the same quantity: no circular wrap). */
int dsa_acorr_fwd(const void* x, int64_t F, int32_t L, int32_t M, int32_t out_format, int32_t dtype,
                  void* r, void* stream);
int dsa_acorr_bwd(const void* gr, const void* x, int64_t F, int32_t L, int32_t M, int32_t out_format,
                  int32_t dtype, void* gx, void* stream);

/* ------------------------------------------------------------------ a12  Levinson-Durbin
 * LevinsonDurbin._forward, levdur.py:113-127: a = solve(toeplitz(r[:M]) + eps I, -r[1:]),
 * K = sqrt(sum r[1:] a + r[0]); out:(F,M+1) = [K, a].  Solved by the Levinson recursion on
 * (r0+eps, r1..rM) (the same system; float64 accumulation inside). */
int dsa_levdur_fwd(const void* r, int64_t F, int32_t M, double eps, int32_t dtype, void* out,
                   void* stream);
int dsa_levdur_bwd(const void* gout, const void* r, const void* out, int64_t F, int32_t M, double eps,
                   int32_t dtype, void* gr, void* stream);

/* ------------------------------------------------------------------ a13  LPC
 * LinearPredictiveCodingAnalysis._forward, lpc.py:137-139 = levdur(acorr(x)); x:(F,L) frames. */
int dsa_lpc_fwd(const void* x, int64_t F, int32_t L, int32_t M, double eps, int32_t dtype, void* scratch,
                void* out, void* stream);
int dsa_lpc_bwd(const void* gout, const void* x, const void* out, int64_t F, int32_t L, int32_t M,
                double eps, int32_t dtype, void* gx, void* stream);
/* Fused LPC branch of the README (README.md:198-201): LPC(Window(Frame(x))) in one kernel.
 * x:(B,T), w:(L) or NULL (a window of ones) -> out:(B,N,M+1).  scratch: see Conventions (the tuned kernel for
 * float32 / lpc_order 24 hands out work items through a counter: the first word, which the kernel leaves zeroed).  pad_mode may carry
 * DSA_LPC_SCRATCH_IS_CLEAN: the caller guarantees that the scratch is zero on entry and used by one call at a time (a buffer per
 * stream that only calls with this flag ever touch) -- the library then skips its fill launch, as DSA_ALGO_SCRATCH_IS_CLEAN does. */
#define DSA_LPC_SCRATCH_IS_CLEAN 0x100
int dsa_frame_window_lpc_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P,
                             const void* w, int32_t center, int32_t pad_mode, int32_t M, double eps,
                             int32_t dtype, void* scratch, void* out, void* stream);
/* pad_mode may also carry DSA_LPC_EXACT_LAGSUMS: the tuned float32 kernel then forms its lag sums as exact float64 sums on the
 * vector unit (the kernel of rounds 1-3; ~2.7 x the launch time) instead of 3-term binary16 splits on the matrix pipe (~1e-7 r[0]
 * from the exact sums: what the reference's own float32 FFT route has).  For near-singular frames (a sinusoid plus tiny noise, small
 * eps) whose Toeplitz system amplifies that error.  (The environment variable DSA_LPC_LAGSUMS=f64 still selects it process-wide.) */
#define DSA_LPC_EXACT_LAGSUMS 0x200
/* Its backward in ONE launch -- the adjoint of frame.py:120-141, window.py:185-193, acorr.py:110-120 and levdur.py:113-127
 * composed (README.md:198-201 with a gradient): gout:(B,N,M+1), x:(B,T), w:(L) or NULL -> gx:(B,T), every element written; no
 * (B N, L) tensor in memory either way.  Covers float32, lpc_order 24, 25 <= frame_length <= 512, constant padding and the
 * (frame_length, frame_period) pairs whose overlap fits a wave's LDS stretch; anything else returns DSA_ERR_UNSUPPORTED and the
 * caller composes dsa_lpc_bwd / dsa_window_bwd / dsa_frame_bwd.  A fixed (non-learnable) window: no window gradient. */
int dsa_frame_window_lpc_bwd(const void* gout, const void* x, int64_t B, int64_t T, int32_t L, int32_t P,
                             const void* w, int32_t center, int32_t pad_mode, int32_t M, double eps,
                             int32_t dtype, void* gx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSPTK_AMD_H */
