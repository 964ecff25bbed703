// Ablation microbenchmark of the tuned STFT kernel (dev tool, not shipped):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value \
//         -I. tools/bench_stft.cpp -o /tmp/bench_stft && /tmp/bench_stft [B]
#include "../diffsptk_amd/csrc/stft.hip"

#include <cmath>
#include <cstdio>
#include <vector>

template <int ABL>
static float run(const float* x, long B, long T, const float* w, const float* tw, float* y, int iters, int waves_per_cu)
{
    const int L = 400, P = 80;
    long N = (T - 1) / P + 1;
    int in_floats = 0;
    int lds = dsa::stft512_lds_bytes(L, P, &in_floats);
    int chunks_per_utt = (int)((N + dsa::kFPW - 1) / dsa::kFPW);
    long total_chunks = B * chunks_per_utt;
    long grid = 256L * waves_per_cu;
    if (grid > total_chunks) grid = total_chunks;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        dsa::stft512_launch<ABL>(false, dim3((unsigned)grid), lds, 0, x, T, N, L, P, 200, 0, w, tw, 1e-9f, 0, 0.f, 3, y,
                                 total_chunks, chunks_per_utt, in_floats);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i)
        dsa::stft512_launch<ABL>(false, dim3((unsigned)grid), lds, 0, x, T, N, L, P, 200, 0, w, tw, 1e-9f, 0, 0.f, 3, y,
                                 total_chunks, chunks_per_utt, in_floats);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}

int main(int argc, char** argv)
{
    long B = argc > 1 ? atol(argv[1]) : 1024, T = 16000;
    long N = (T - 1) / 80 + 1;
    std::vector<float> hx(B * T), hw(400), htw(1024);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < 400; ++i) hw[i] = 0.42f - 0.5f * cosf(2 * M_PI * i / 399) + 0.08f * cosf(4 * M_PI * i / 399);
    for (int m = 0; m < 512; ++m) {
        htw[2 * m] = cosf(2 * M_PI * m / 512);
        htw[2 * m + 1] = -sinf(2 * M_PI * m / 512);
    }
    float *x, *w, *tw, *y;
    hipMalloc(&x, B * T * 4);
    hipMalloc(&w, 400 * 4);
    hipMalloc(&tw, 1024 * 4);
    hipMalloc(&y, B * N * 257 * 4);
    hipMemcpy(x, hx.data(), B * T * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), 400 * 4, hipMemcpyHostToDevice);
    hipMemcpy(tw, htw.data(), 1024 * 4, hipMemcpyHostToDevice);
    double bytes = (double)B * N * 1348.0;
    for (int wpc : {8, 12, 16}) {
        float t0 = run<0>(x, B, T, w, tw, y, 20, wpc);
        float t1 = run<1>(x, B, T, w, tw, y, 20, wpc);
        float t2 = run<2>(x, B, T, w, tw, y, 20, wpc);
        float t3 = run<3>(x, B, T, w, tw, y, 20, wpc);
#ifdef DSA_STFT_TIMING
        {
            run<0>(x, B, T, w, tw, y, 1, wpc);
            unsigned long long st[16];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_stft_stamps), sizeof(st));
            printf("  phase cycles: stage %llu | window %llu | fft1 %llu | tw+T-write %llu | T-read %llu | fft2 %llu | Z-write %llu | split %llu | copy-out %llu\n",
                   st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6],
                   st[8] - st[7], st[9] - st[8]);
        }
#endif
        printf("B=%ld waves/CU=%2d  full %8.1f us (%6.1f GB/s) | no-store %8.1f | no-fft %8.1f | no-load %8.1f\n", B, wpc, t0,
               bytes / t0 * 1e-3, t1, t2, t3);
    }
    return 0;
}
