// Which physical SIMD does a wave run on?  Records HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20) of every wave of a grid of
// single-wave blocks that hold ~160 VGPRs each (3 waves per SIMD, like the gated solve), and prints how the waves spread.
//   hipcc -O3 --offload-arch=gfx950 -o hwid_probe hwid_probe.hip && ./hwid_probe [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void __launch_bounds__(64, 3) k_probe(unsigned *out, double *sink, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    // hold many registers so that only 3 waves fit a SIMD, and stay resident for a while
    double acc[72];
#pragma unroll
    for (int i = 0; i < 72; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int s = 0; s < spin; ++s) {
#pragma unroll
        for (int i = 0; i < 72; ++i) acc[i] = __builtin_fma(acc[i], 1.0000001, acc[(i + 1) % 72] * 1e-9);
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 72; ++i) t += acc[i];
    if (t == 12345.678) sink[0] = t;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 3072;
    const int spin = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *d; double *sink;
    hipMalloc(&d, blocks * 2 * sizeof(unsigned)); hipMalloc(&sink, 8);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64), 0, 0, d, sink, spin);
    std::vector<unsigned> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> per_simd;
    std::set<unsigned> xccs, ses, shs, cus, simds, pipes, waves;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
        const unsigned wave = hw & 0xF, simd = (hw >> 4) & 3, pipe = (hw >> 6) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        xccs.insert(xcc); ses.insert(se); shs.insert(sh); cus.insert(cu); simds.insert(simd); pipes.insert(pipe); waves.insert(wave);
        per_simd[((unsigned long long)xcc << 16) | (se << 13) | (sh << 12) | (cu << 8) | (simd << 4)]++;
        if (b < 16) printf("block %4d hw %08x xcc %08x -> xcc %u se %u sh %u cu %2u simd %u wave %u pipe %u\n", b, hw, h[2 * b + 1], xcc, se, sh, cu, simd, wave, pipe);
    }
    std::map<int, int> hist;
    for (auto &kv : per_simd) hist[kv.second]++;
    printf("blocks %d: distinct SIMDs %zu; xcc %zu se %zu sh %zu cu %zu simd %zu wave-ids %zu\n", blocks, per_simd.size(), xccs.size(), ses.size(), shs.size(), cus.size(), simds.size(), waves.size());
    printf("cu ids:"); for (unsigned c : cus) printf(" %u", c); printf("\nse ids:"); for (unsigned c : ses) printf(" %u", c); printf("\n");
    for (auto &kv : hist) printf("  %d SIMDs hold %d waves\n", kv.second, kv.first);
    return 0;
}
