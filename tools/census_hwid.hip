// Census of where the dispatcher puts the workgroups of a band-kernel-shaped launch (512 threads, ~70 KB of LDS: two per CU):
// per workgroup HW_REG_HW_ID fields (CU, SE, SH, TG_ID = the workgroup's slot on its CU), HW_REG_XCC_ID, start / end clock.
// Speed-only knowledge (the stagger option of conv_band_f16s3.hip): results never depend on it.
//   hipcc --offload-arch=gfx950 -O2 -o tools/census_hwid tools/census_hwid.hip && tools/census_hwid [blocks] [spin_us]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };
__global__ __launch_bounds__(512) void census(Rec* out, int spin_ticks) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) {
        Rec r;
        r.hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, all 32 bits
        r.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));        // HW_REG_XCC_ID[3:0]
        r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x] = r;
    }
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 722, spin_us = argc > 2 ? atoi(argv[2]) : 20;
    // optional CU mask (hipExtStreamCreateWithCUMask): argv[3] = 8 hex words, e.g. ffffffff,ffffffff,ffffffff,ffffffff,0,0,0,0
    hipStream_t stream = 0;
    if (argc > 3) {
        unsigned mask[8] = {0}; int n = 0; char* tok = strtok(argv[3], ",");
        while (tok && n < 8) { mask[n++] = (unsigned)strtoul(tok, nullptr, 16); tok = strtok(nullptr, ","); }
        hipError_t e = hipExtStreamCreateWithCUMask(&stream, 8, mask);
        printf("hipExtStreamCreateWithCUMask -> %s\n", hipGetErrorString(e));
        if (e != hipSuccess) return 1;
    }
    Rec* d; hipMalloc(&d, blocks * sizeof(Rec));
    hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(census, dim3(blocks), dim3(512), 70 * 1024, stream, d, spin_us * 100);
        hipDeviceSynchronize();
    }
    std::vector<Rec> h(blocks); hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull; for (auto& r : h) if (r.t0 < tmin) tmin = r.t0;
    std::map<unsigned, std::vector<int>> per_cu;      // (xcc, se, sh, cu) -> blocks
    int tg_hist[16] = {0}, first_round_tg[16] = {0};
    for (int b = 0; b < blocks; ++b) {
        const unsigned w = h[b].hwid, cu = (w >> 8) & 15, sh = (w >> 12) & 1, se = (w >> 13) & 7, tg = (w >> 16) & 15;
        per_cu[(h[b].xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        ++tg_hist[tg];
        if ((h[b].t0 - tmin) < 200) ++first_round_tg[tg];
        if (b < 24 || b >= blocks - 8) printf("block %4d xcc %u se %u sh %u cu %2u tg %2u simd %u wave %u start %6.2f us end %6.2f us\n", b, h[b].xcc, se, sh, cu, tg, (w >> 4) & 3, w & 15,
                                              (h[b].t0 - tmin) / 100.0, (h[b].t1 - tmin) / 100.0);
    }
    printf("distinct CUs %zu\n", per_cu.size());
    { int per_xcc[8] = {0}; for (auto& kv : per_cu) ++per_xcc[(kv.first >> 12) & 7]; printf("CUs per XCC:"); for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]); printf("\n"); }
    printf("TG_ID histogram (all):"); for (int i = 0; i < 16; ++i) printf(" %d", tg_hist[i]); printf("\n");
    printf("TG_ID histogram (started within 2 us):"); for (int i = 0; i < 16; ++i) printf(" %d", first_round_tg[i]); printf("\n");
    int shown = 0;
    for (auto& kv : per_cu) if (shown++ < 6) {
        printf("cu key %05x:", kv.first);
        for (int b : kv.second) printf(" [b%d tg%u %.1f-%.1f]", b, (h[b].hwid >> 16) & 15, (h[b].t0 - tmin) / 100.0, (h[b].t1 - tmin) / 100.0);
        printf("\n");
    }
    // co-resident pairs of the first round: do their TG_ID parities differ?
    int pairs = 0, differ = 0;
    for (auto& kv : per_cu) {
        std::vector<int> fr; for (int b : kv.second) if ((h[b].t0 - tmin) < 200) fr.push_back(b);
        if (fr.size() == 2) { ++pairs; if ((((h[fr[0]].hwid ^ h[fr[1]].hwid) >> 16) & 1)) ++differ; }
    }
    printf("first-round pairs %d, with different TG_ID parity %d\n", pairs, differ);
    return 0;
}
