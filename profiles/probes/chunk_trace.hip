// Tuning probe: a timeline of the gated solve.  Builds the kernels with -DRP_TRACE (every chunk records which SIMD ran it and
// when, in 100 MHz ticks) and prints one line per chunk of the benchmark batch's second solve:
//     chunk xcc se cu simd start_ticks end_ticks steps
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DRP_TRACE -I ../../include -o chunk_trace chunk_trace.hip \
//         ../../rocket_path_amd/csrc/schedule.hip ../../rocket_path_amd/csrc/rp_batch.cpp
//   ./chunk_trace > trace.txt && python chunk_trace_analyze.py shipped=trace.txt
#include "../../rocket_path_amd/csrc/ip_kernels.hip"

#include <cstdio>
#include <vector>

static unsigned long long mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double u01(unsigned long long seed, unsigned long long ctr) { return (double)(mix(seed + ctr) >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : (size_t)1 << 20;
    std::vector<double> p0(n), p1(n), p2(n);
    for (size_t i = 0; i < n; ++i) {      // rocket_path_amd/problems.py, DIST_MONOTONE, seed 12345
        p0[i] = 1000.0 * u01(12345, 3 * i + 1);
        p1[i] = p0[i] + 10.0 + 500.0 * u01(12345, 3 * i + 2);
        p2[i] = p1[i] + 10.0 + 500.0 * u01(12345, 3 * i + 3);
    }
    rp_batch *b = nullptr;
    if (rp_batch_create(&b, RP_VARIANT_F3, RP_DTYPE_F64, n, 0, nullptr) != RP_OK) { fprintf(stderr, "%s\n", rp_last_error()); return 1; }
    for (int rep = 0; rep < 2; ++rep) {
        rp_batch_set_problems(b, p0.data(), p1.data(), p2.data());
        rp_batch_solve(b, 1e-8, 200, 0);
        rp_batch_sync(b);
    }
    rp_reduction red;
    rp_batch_reduce(b, &red);
    const size_t chunks = (n + 63) / 64;
    std::vector<unsigned long long> t(4 * 32768);
    if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(rp::g_trace), t.size() * sizeof(unsigned long long)) != hipSuccess) { fprintf(stderr, "no trace\n"); return 1; }
    fprintf(stderr, "steps %.0f converged %.0f\n", red.total_steps, red.n_converged);
    for (size_t c = 0; c < chunks && c < 32768; ++c) {
        const unsigned hw = (unsigned)t[4 * c], xcc = (unsigned)(t[4 * c] >> 32);
        printf("%zu %u %u %u %u %llu %llu %llu\n", c, xcc, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, t[4 * c + 1], t[4 * c + 2], t[4 * c + 3]);
    }
    rp_batch_destroy(b);
    return 0;
}
