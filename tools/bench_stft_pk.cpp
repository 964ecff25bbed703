// Ablation / timeline microbenchmark of the packed STFT forward kernel (dev tool, not shipped):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value \
//         -DDSA_STFT_TIMING -Xclang -target-feature -Xclang -packed-fp32-ops -I. tools/bench_stft_pk.cpp -o build/bench_stft_pk
#include "../diffsptk_amd/csrc/stft.hip"

#include <cmath>
#include <cstdio>
#include <vector>

static const float *gx, *gw, *gtw;
static float* gy;
static long gB, gT = 16000, gN;

template <int ABL, bool DIRECT = false, bool PF2 = false>
static float run(int iters, int waves_per_cu)
{
    const int L = 400, P = 80;
    int chunks_per_utt = (int)((gN + dsa::kFPW - 1) / dsa::kFPW);
    long total_chunks = gB * chunks_per_utt;
    long grid = 256L * waves_per_cu;
    if (grid > total_chunks) grid = total_chunks;
    const int lds2 = dsa::stft512_lds_bytes2();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&] {
        if (PF2)
            hipLaunchKernelGGL((dsa::stft512_fwd_pk_kernel<ABL, 400, true, 0, true>), dim3((unsigned)((grid + 3) / 4)), dim3(256),
                               4 * dsa::kFPW * dsa::kZS * 8 + 256 * 8 + 16 * 13 * 8 + 64, 0, gx, gT, gN, L, P, 200, gw, gtw, 1e-9f, gy,
                               total_chunks, chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, 0, 0);
        else
        hipLaunchKernelGGL((dsa::stft512_fwd_pk_kernel<ABL, 400, DIRECT>), dim3((unsigned)((grid + 1) / 2)), dim3(128), lds2, 0, gx, gT, gN, L,
                           P, 200, gw, gtw, 1e-9f, gy, total_chunks, chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, 0, 0);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}

int main(int argc, char** argv)
{
    gB = argc > 1 ? atol(argv[1]) : 1024;
    gN = (gT - 1) / 80 + 1;
    std::vector<float> hx(gB * gT), hw(400), htw(1024);
    for (auto& v : hx) v = (float)(rand() % 65536) / 65536.f - 0.5f;
    for (int i = 0; i < 400; ++i) hw[i] = 0.42f - 0.5f * cosf(2 * M_PI * i / 399) + 0.08f * cosf(4 * M_PI * i / 399);
    for (int m = 0; m < 512; ++m) {
        htw[2 * m] = cosf(2 * M_PI * m / 512);
        htw[2 * m + 1] = -sinf(2 * M_PI * m / 512);
    }
    float *x, *w, *tw;
    hipMalloc(&x, gB * gT * 4);
    hipMalloc(&w, 400 * 4);
    hipMalloc(&tw, 1024 * 4);
    hipMalloc(&gy, gB * gN * 257 * 4);
    hipMemcpy(x, hx.data(), gB * gT * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), 400 * 4, hipMemcpyHostToDevice);
    hipMemcpy(tw, htw.data(), 1024 * 4, hipMemcpyHostToDevice);
    gx = x, gw = w, gtw = tw;
    double bytes = (double)gB * gN * 1348.0;
    for (int rep = 0; rep < 2; ++rep) {   // twice: the first round runs while the clocks ramp
        const int wpc = 16;
        printf("B=%ld  full %.1f us (%.0f GB/s)\n", gB, run<0>(20, wpc), bytes / run<0>(20, wpc) * 1e-3);
        printf("  no-store(1) %.1f | no-fft(2) %.1f | no-load(4) %.1f | no-twtab(8) %.1f | no-transpose(16) %.1f | no-Zrt(32) %.1f | no-stage(64) %.1f\n",
               run<1>(20, wpc), run<2>(20, wpc), run<4>(20, wpc), run<8>(20, wpc), run<16 + 8>(20, wpc), run<32>(20, wpc), run<64>(20, wpc));
        printf("  no-lds-at-all(4+8+16+32+64) %.1f | +no-fft %.1f | only loads+stage+stores(2+8+16+32) %.1f | no load/store(1+4) %.1f | no-store,no-load,no-fft %.1f\n",
               run<4 + 8 + 16 + 32 + 64>(20, wpc), run<2 + 4 + 8 + 16 + 32 + 64>(20, wpc), run<2 + 8 + 16 + 32>(20, wpc), run<1 + 4>(20, wpc),
               run<1 + 2 + 4>(20, wpc));
        for (int wpc2 : {8, 12}) printf("  waves/CU=%d full %.1f\n", wpc2, run<0>(20, wpc2));
        printf("  DIRECT stores: full %.1f | no-store %.1f | no-load %.1f | no-fft %.1f | no load/store %.1f | wpc12 %.1f\n", run<0, true>(20, wpc), run<1, true>(20, wpc),
               run<4, true>(20, wpc), run<2, true>(20, wpc), run<5, true>(20, wpc), run<0, true>(20, 12));
    }
    for (int rep = 0; rep < 3; ++rep)
        printf("DIRECT, twiddle-table reads: base %.1f | no-twtab(8) %.1f | base %.1f | no-twtab(8) %.1f\n", run<0, true>(20, 16), run<8, true>(20, 16),
               run<0, true>(20, 16), run<8, true>(20, 16));
    for (int rep = 0; rep < 2; ++rep)
        printf("DIRECT variants: base %.1f | xcd-contiguous %.1f | nontemporal %.1f | both %.1f | xcd, staged %.1f\n", run<0, true>(20, 16), run<256, true>(20, 16),
               run<512, true>(20, 16), run<768, true>(20, 16), run<256, false>(20, 16));
    for (int rep = 0; rep < 3; ++rep)
        printf("DIRECT, waves per CU: 16 %.1f | 14 %.1f | 12 %.1f | 10 %.1f | 16 %.1f | 14 %.1f\n", run<0, true>(20, 16), run<0, true>(20, 14), run<0, true>(20, 12),
               run<0, true>(20, 10), run<0, true>(20, 16), run<0, true>(20, 14));
    for (int rep = 0; rep < 3; ++rep)
        printf("DIRECT, XCD-chunked workgroup order: base %.1f | C=2 %.1f | C=4 %.1f | C=8 %.1f | base %.1f\n", run<0, true>(20, 16), run<2048, true>(20, 16),
               run<1024, true>(20, 16), run<4096, true>(20, 16), run<0, true>(20, 16));
    for (int rep = 0; rep < 3; ++rep)
        printf("two-pass-ahead fetch (PF2): base %.1f | PF2 %.1f | base %.1f | PF2 %.1f | PF2 no-store %.1f | PF2 no-load %.1f\n", run<0, true>(20, 16),
               run<0, true, true>(20, 16), run<0, true>(20, 16), run<0, true, true>(20, 16), run<1, true, true>(20, 16), run<4, true, true>(20, 16));
    {   // PF2 against DIRECT, element by element
        size_t n = (size_t)gB * gN * 257;
        std::vector<float> a(n), b(n);
        hipMemset(gy, 0, n * 4);
        run<0, true>(1, 16);
        hipMemcpy(a.data(), gy, n * 4, hipMemcpyDeviceToHost);
        hipMemset(gy, 0, n * 4);
        run<0, true, true>(1, 16);
        hipMemcpy(b.data(), gy, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
        printf("PF2 vs DIRECT: %zu differing of %zu\n", bad, n);
    }
    {   // DIRECT against the staged variant, element by element
        size_t n = (size_t)gB * gN * 257;
        std::vector<float> a(n), b(n);
        hipMemset(gy, 0, n * 4);
        run<0>(1, 16);
        hipMemcpy(a.data(), gy, n * 4, hipMemcpyDeviceToHost);
        hipMemset(gy, 0, n * 4);
        run<0, true>(1, 16);
        hipMemcpy(b.data(), gy, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        double worst = 0;
        for (size_t i = 0; i < n; ++i) {
            double d = fabs((double)a[i] - b[i]);
            if (d > 0) ++bad;
            if (d > worst) worst = d;
        }
        printf("DIRECT vs staged: %zu differing of %zu, worst %.3e\n", bad, n, worst);
    }
#ifdef DSA_STFT_TIMING
    auto timeline = [&](const char* tag, float t) {
        unsigned long long st[64];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(dsa::g_stft_pk_stamps), sizeof(st));
        printf("timeline of wave 0, %s (cycles; kernel incl. launch %.1f us): prologue %llu | whole wave %llu\n", tag, t, st[1] - st[0], st[2] - st[0]);
        printf("  prologue: to prefetch issued %llu | tables loaded %llu | first stretch staged %llu | second prefetch issued %llu\n", st[3] - st[0],
               st[4] - st[3], st[5] - st[4], st[1] - st[5]);
        printf("  pass starts (delta):");
        for (int i = 1; i < 14; ++i) printf(" %llu", st[8 + i] > st[8 + i - 1] ? st[8 + i] - st[8 + i - 1] : 0ULL);
        printf("\n  pass 3 phases: window-read %llu | fft1 %llu | tw+T-write %llu | T-read %llu | fft2 %llu | Z-write %llu | pair-read %llu | "
               "split(+fetch wait, stores) %llu | copy-out %llu | next stretch staged + prefetch issued %llu\n",
               st[41] - st[8 + 3], st[42] - st[41], st[43] - st[42], st[44] - st[43], st[45] - st[44], st[46] - st[45],
               st[47] - st[46], st[48] - st[47], st[49] - st[48], st[40] - st[49]);
    };
    timeline("staged", run<128>(1, 16));
    timeline("DIRECT", run<128, true>(1, 16));
    timeline("DIRECT, no stores", run<128 + 1, true>(1, 16));
    timeline("DIRECT, no loads", run<128 + 4, true>(1, 16));
#endif
    return 0;
}
