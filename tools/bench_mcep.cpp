// Phase-timing microbenchmark of the MFMA mel-cepstral kernel (dev tool, not shipped):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value -DDSA_MCEP_TIMING \
//         tools/bench_mcep.cpp -o build/bench_mcep && build/bench_mcep
#include "../diffsptk_amd/csrc/mcep_mfma.hip"

#include <cmath>
#include <cstdio>
#include <vector>

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
    hipMalloc(&X, hX.size() * 4); hipMalloc(&G, hG.size() * 4); hipMalloc(&D, hD.size() * 4);
    hipMalloc(&E, hE.size() * 4); hipMalloc(&av, 100); hipMalloc(&mc, F * M1 * 4);
    hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(E, hE.data(), hE.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(av, hav.data(), 100, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant : {4, 8, 16}) {
        float ms = 0;
        for (int rep = 0; rep < (argc > 2 ? atoi(argv[2]) : 3); ++rep) {
            hipEventRecord(e0);
            if (variant == 16) dsa::launch_h<8>(X, F, 10, G, D, E, av, mc, nullptr, 0, "h8");
            else if (variant == 4) dsa::launch_v2<4>(X, F, 10, G, D, E, av, mc, nullptr, 0, "w4");
            else dsa::launch_v2<8>(X, F, 10, G, D, E, av, mc, nullptr, 0, "w8");
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long st[64];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_mcep_stamps), sizeof(st));
        printf("   wave 0: %llu ticks from start to last tile end, %llu tiles (all reps) => tick rate %.1f MHz if the wave spans the kernel\n",
               st[9] - st[8], st[10], (st[9] - st[8]) / (ms * 1e3));
        printf("waves=%d  kernel %.3f ms | cycles: mfma+exp %llu  rt->lds %llu  build %llu  elim %llu  backsub %llu\n", variant,
               ms, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4]);
        if (variant == 16) {
            printf("   wave 0 tile starts (ticks since kernel start: tile id):");
            for (int i = 0; i < 12 && st[32 + i] > st[8]; ++i) printf("  %llu:%llu", st[32 + i] - st[8], st[48 + i]);
            printf("  end %llu\n", st[9] - st[8]);
        }
        if (variant == 16)
            printf("   third tile of wave 0: X load + log2 %llu  mc0 chain %llu  10 iterations %llu  store %llu  queue %llu\n",
                   st[17] - st[16], st[18] - st[17], st[19] - st[18], st[20] - st[19], st[21] - st[20]);
    }
    std::vector<float> h(8);
    hipMemcpy(h.data(), mc, 32, hipMemcpyDeviceToHost);
    printf("mc[0..3] = %g %g %g %g\n", h[0], h[1], h[2], h[3]);
    return 0;
}
