// Phase timing of the matrix-core filter-bank kernel (dev tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DDSA_FBANK_TIMING -Iinclude tools/bench_fbank.cpp -o build/bench_fbank
#include "../diffsptk_amd/csrc/fbank.hip"

#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char** argv)
{
    long F = argc > 1 ? atol(argv[1]) : 204800;
    const int K = 257, C = 40;
    std::vector<float> hx(F * K), hH((size_t)K * C, 0.f);
    srand(1);
    for (auto& v : hx) v = 0.5f + (float)(rand() % 65536) / 65536.f;
    for (int c = 0; c < C; ++c) {  // triangles on a warped axis
        double lo = 256.0 * pow((double)c / (C + 1), 1.8), mid = 256.0 * pow((double)(c + 1) / (C + 1), 1.8),
               hi = 256.0 * pow((double)(c + 2) / (C + 1), 1.8);
        for (int k = 1; k < K; ++k) {
            double v = k < mid ? (k - lo) / (mid - lo) : (hi - k) / (hi - mid);
            if (v > 0) hH[(size_t)k * C + c] = (float)v;
        }
    }
    float *x, *H, *y, *E;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&H, hH.size() * 4); hipMalloc(&y, F * C * 4); hipMalloc(&E, F * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(H, hH.data(), hH.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int up = 1; up >= 0; --up) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) dsa_fbank_fwd(x, F, K, H, C, 1e-5, 0.0, up, DSA_F32, y, E, nullptr);
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) dsa_fbank_fwd(x, F, K, H, C, 1e-5, 0.0, up, DSA_F32, y, E, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_fbank_stamps), sizeof(st));
        printf("use_power=%d F=%ld: %.1f us (%.0f GB/s) | prologue %llu | tile: stage-write %llu  issue %llu  products %llu  epilogue %llu | from kernel start to 2nd tile %llu | prologue parts: first issue %llu  H copy %llu  masks %llu  compaction %llu  images %llu\n",
               up, F, ms * 100, (double)F * (K + C + 1) * 4 / (ms / 10) * 1e-6, st[1] - st[0], st[3] - st[2], st[4] - st[3],
               st[5] - st[4], st[6] - st[5], st[2] - st[0], st[7] - st[0], st[8] - st[7], st[9] - st[8], st[10] - st[9], st[11] - st[10]);
    }
    return 0;
}
