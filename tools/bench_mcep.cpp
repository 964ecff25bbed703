// Phase timing of the tuned mel-cepstral forward kernel (dev tool, not shipped): cycle stamps of wave 0 in the second Newton
// step of its first tile (every other wave of the chip is in its steady state by then).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value -ffp-contract=on -DDSA_MCEP_TIMING \
//         -Iinclude tools/bench_mcep.cpp -o build/bench_mcep [-DDSA_MCEP_SOLVE_VALU ...] && build/bench_mcep
#include "../diffsptk_amd/csrc/mcep_mfma.hip"

#include <cmath>
#include <cstdio>
#include <vector>

namespace dsa {   // the two symbols of common.h's error plumbing that live in another translation unit of the library
thread_local char g_last_error[256];
thread_local const char* g_last_kernel = "";
int thsolve_fix_marked(const void*, const void*, const void*, int64_t, int, void*, hipStream_t, int, int, const void*) { return 0; }   // csrc/mgc.hip in the library
}

int main(int argc, char** argv)
{
    long F = argc > 1 ? atol(argv[1]) : 204800;
    const int K = 257, M1 = 25, M2 = 49;
    std::vector<float> hX(F * K), hG(K * M1), hD(M1 * K), hE(K * M2), hav(M1);
    srand(1);
    for (auto& v : hX) v = 0.5f + (float)rand() / RAND_MAX;
    // benign well-conditioned stand-ins for the tables: timing only, values irrelevant
    for (int k = 0; k < K; ++k)
        for (int m = 0; m < M1; ++m) {
            hG[k * M1 + m] = (m == 0 ? 1.f : 0.1f) / K * cosf(0.01f * k * m);
            hD[m * K + k] = cosf(0.012f * k * m) * (m == 0 ? 1.f : 0.05f);
        }
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < M2; ++j) hE[k * M2 + j] = cosf(3.14159265f * k * j / 256.f) / 512.f * (k == 0 || k == 256 ? 1.f : 2.f);
    for (int m = 0; m < M1; ++m) hav[m] = powf(-0.42f, (float)m);
    float *X, *G, *D, *E, *av, *mc;
    void *img, *scratch;
    hipMalloc(&X, hX.size() * 4); hipMalloc(&G, hG.size() * 4); hipMalloc(&D, hD.size() * 4);
    hipMalloc(&E, hE.size() * 4); hipMalloc(&av, 128); hipMalloc(&mc, F * M1 * 4);
    hipMalloc(&img, dsa::mcep_mfma_images_bytes()); hipMalloc(&scratch, 256);
    hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(E, hE.data(), hE.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(av, hav.data(), 100, hipMemcpyHostToDevice);
    dsa::mcep_mfma_prepare(G, D, E, img, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        int rc = dsa::mcep_mfma_fwd(X, F, 10, G, D, E, av, img, scratch, mc, nullptr, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("launch failed %d\n", rc); return 1; }
        if (rep >= reps / 2 && ms < best) best = ms;
    }
    unsigned long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_mcep_stamps), sizeof(st));
    // the step stamps 0 .. 11 are the low 32 bits of the counter: unwrap them relative to stamp 0
    for (int i = 1; i < 12; ++i) st[i] = st[0] + (unsigned)((unsigned)st[i] - (unsigned)st[0]);
    printf("kernel %.4f ms (best of the last %d) | wave 0, tile 0, step 1, cycles: chains+exp %llu  rt->lds %llu  build %llu  elim %llu  backsub+update %llu  = %llu\n",
           best, reps - reps / 2, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[5] - st[0]);
    printf("   inside chains+exp: mc split + Nyquist dot %llu  pass 1 %llu  max reduction %llu  pass 2 %llu  Nyquist + rt[48] %llu\n", st[6] - st[0],
           st[7] - st[6], st[8] - st[7], st[13] - st[12], st[1] - st[9]);
    printf("   third tile of wave 0: X load + log2 %llu  mc0 chain %llu  10 iterations %llu  store %llu  queue %llu\n",
           st[17] - st[16], st[18] - st[17], st[19] - st[18], st[20] - st[19], st[21] - st[20]);
    printf("   wave 0: %llu ticks from kernel entry to its last tile's end; tile starts (ticks since entry: tile id):", st[13] - st[12]);
    for (int i = 0; i < 12 && st[32 + i] > st[12]; ++i) printf("  %llu:%llu", st[32 + i] - st[12], st[48 + i]);
    printf("\n   => %.3f GHz if wave 0 spans the kernel\n", (st[13] - st[12]) / (best * 1e6));
    {   // how the launch ends: per wave slot, entry / exit on the 100 MHz clock and the tiles it ran
        std::vector<unsigned long long> sl(2048 * 3);
        hipMemcpyFromSymbol(sl.data(), HIP_SYMBOL(dsa::g_mcep_slotlog), sl.size() * 8);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < 2048; ++i) { if (sl[3 * i] && sl[3 * i] < t0) t0 = sl[3 * i]; if (sl[3 * i + 1] > t1) t1 = sl[3 * i + 1]; }
        int hist_first[16] = {0}, hist_second[16] = {0}, hist_end[10] = {0};
        for (int i = 0; i < 2048; ++i) {
            if (!sl[3 * i]) continue;
            const int nt = sl[3 * i + 2] < 15 ? (int)sl[3 * i + 2] : 15;
            ((i & 7) < 4 ? hist_first : hist_second)[nt]++;   // waves 0..3 of a workgroup: the older wave of each SIMD pair
            hist_end[(sl[3 * i + 1] - t0) * 10 / (t1 - t0 + 1)]++;
        }
        printf("   slots: first entry -> last exit %.1f us; tiles per slot, older wave of a SIMD pair:", (t1 - t0) / 100.0);
        for (int i = 0; i < 16; ++i) if (hist_first[i]) printf(" %d:%d", i, hist_first[i]);
        printf("  younger wave:");
        for (int i = 0; i < 16; ++i) if (hist_second[i]) printf(" %d:%d", i, hist_second[i]);
        printf("\n   slot exits by tenth of the launch:");
        for (int i = 0; i < 10; ++i) printf(" %d", hist_end[i]);
        printf("\n");
    }
    std::vector<float> h(8);
    hipMemcpy(h.data(), mc, 32, hipMemcpyDeviceToHost);
    printf("   mc[0..3] = %g %g %g %g\n", h[0], h[1], h[2], h[3]);
    return 0;
}
