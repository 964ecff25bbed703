// Phase timing of the tuned mel-cepstral backward kernel (dev tool, not shipped): cycle stamps of wave 0 in the second step of
// the reverse sweep over its first tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value -ffp-contract=on -DDSA_MCEP_TIMING \
//         -Iinclude tools/bench_mcep_bwd.cpp -o build/bench_mcep_bwd && build/bench_mcep_bwd
#include "../diffsptk_amd/csrc/mcep_mfma.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dsa {
thread_local char g_last_error[256];
thread_local const char* g_last_kernel = "";
int thsolve_fix_marked(const void*, const void*, const void*, int64_t, int, void*, hipStream_t, int, int, const void*) { return 0; }   // csrc/mgc.hip in the library
}

int main(int argc, char** argv)
{
    long F = argc > 1 ? atol(argv[1]) : 204800;
    const int K = 257, M1 = 25, M2 = 49, NIT = 10;
    std::vector<float> hX(F * K), hG(K * M1), hD(M1 * K), hE(K * M2), hav(M1), hg(F * M1);
    srand(1);
    for (auto& v : hX) v = 0.5f + (float)rand() / RAND_MAX;
    for (auto& v : hg) v = (float)rand() / RAND_MAX - 0.5f;
    for (int k = 0; k < K; ++k)
        for (int m = 0; m < M1; ++m) {
            hG[k * M1 + m] = (m == 0 ? 1.f : 0.1f) / K * cosf(0.01f * k * m);
            hD[m * K + k] = cosf(0.012f * k * m) * (m == 0 ? 1.f : 0.05f);
        }
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < M2; ++j) hE[k * M2 + j] = cosf(3.14159265f * k * j / 256.f) / 512.f * (k == 0 || k == 256 ? 1.f : 2.f);
    for (int m = 0; m < M1; ++m) hav[m] = powf(-0.42f, (float)m);
    float *X, *G, *D, *E, *av, *mc, *hist, *gmc, *gX;
    void *img, *scratch;
    hipMalloc(&X, hX.size() * 4); hipMalloc(&G, hG.size() * 4); hipMalloc(&D, hD.size() * 4);
    hipMalloc(&E, hE.size() * 4); hipMalloc(&av, 128); hipMalloc(&mc, F * M1 * 4); hipMalloc(&hist, (size_t)(NIT + 1) * F * M1 * 4 + (size_t)NIT * F * M2 * 4);
    hipMalloc(&gmc, F * M1 * 4); hipMalloc(&gX, hX.size() * 4);
    hipMalloc(&img, dsa::mcep_mfma_images_bytes()); hipMalloc(&scratch, 256);
    hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(E, hE.data(), hE.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(av, hav.data(), 100, hipMemcpyHostToDevice);
    hipMemcpy(gmc, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    dsa::mcep_mfma_prepare(G, D, E, img, 0);
    const bool rt = argc > 3 ? atoi(argv[3]) != 0 : true;   // the forward saves its rt rows (DSA_ALGO_HIST_HAS_RT), the product's default
    dsa::mcep_mfma_fwd(X, F, NIT, G, D, E, av, img, scratch, mc, hist, 0, false, nullptr, rt);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        int rc = dsa::mcep_mfma_bwd(gmc, X, hist, F, NIT, av, img, scratch, gX, 0, false, rt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("launch failed %d\n", rc); return 1; }
        if (rep >= reps / 2 && ms < best) best = ms;
    }
    unsigned long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_mcep_stamps), sizeof(st));
    unsigned s[16];
    for (int i = 0; i < 16; ++i) s[i] = (unsigned)st[40 + i];
    printf("kernel %.4f ms (best of the last %d) | wave 0, first tile, second step, cycles: forward chains + windows %u  build + solve %u  rtbar + scale %u  ebar chain %u  mbar chain %u  = %u\n",
           best, reps - reps / 2, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[5] - s[0]);
    printf("   forward: first chain %u  reduction + second chain %u  windows %u | mbar: prologue %u  bodies %u  epilogue %u\n", s[6] - s[0], s[7] - s[6],
           s[1] - s[7], s[8] - s[4], s[9] - s[8], s[5] - s[9]);
    if (rt && !(getenv("DSA_MCEP_BWD2") && getenv("DSA_MCEP_BWD2")[0] == '0'))
        printf("   two-wave kernel (stamps as: windows %u  build %u  elimination %u  back substitution %u  rtbar %u  exchange %u | group 0 %u  group 1: products %u  e + zbar %u  mbar %u | groups 2, 3 + Nyquist %u | step %u)\n",
               s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], s[7] - s[6], s[8] - s[7], s[9] - s[8], s[10] - s[9],
               s[11] - s[10], s[11] - s[0]);
    std::vector<float> h(4);
    hipMemcpy(h.data(), gX, 16, hipMemcpyDeviceToHost);
    printf("   gX[0..3] = %g %g %g %g\n", h[0], h[1], h[2], h[3]);
    return 0;
}
