// headless_shell.cpp -- rocket_path.cpp's dispatch without the window.
//
// Same contract as the GLUT shell (rocket_path.cpp:33-75, 130-162): a table of Problem
// objects, init() on all of them at start, onActivate() on the current one, keys forwarded
// to the current problem, F-keys switch problems.  The F3 and F4 slots hold the batched GPU
// problems; F1/F2 (fixptpath, onedpath) are out of scope and left empty.
//
//   rp_headless [--n N] [--seed S] [--f4] [--f32] [--gpus G] [--watch I] [--keys "i n n s"] [--solve]
//   rp_headless --gpus G --n N --seed S --bench K [--warmup W]
//
// --bench K (with --gpus G): the single-process multi-GPU path timed like bench.py times the one-process-per-GPU path --
// W untimed then K timed passes (restart + fused gated solve of every shard, gap < 1e-8, cap 200) plus the one RCCL
// summary all-reduce -- and ONE JSON line with bench.py's keys, per-device HIP-event times included.
//
// Redisplay works as under GLUT: a handled key makes the problem call repaint() (here: the hook below marks the
// window dirty, as glutPostRedisplay does, rocket_path.cpp:178-182), and the shell then calls onDraw() on the current
// problem once (rocket_path.cpp:101-106).  The number of redraws is reported on stderr at exit; the token `p` prints
// what the last onDraw() fetched for the watched problem (--watch I, default 0).
//
// --gpus G (G >= 1) puts a ShardedOneDPathIP (G devices, one process, RCCL summary) in the F3 slot.
//
// --keys takes space-separated tokens: single characters are onKey() (SPACE for ' '),
// F3/F4 switch problems, HOME END PGUP PGDN LEFT RIGHT UP DOWN are special keys, nK repeats
// 'n' K times in one fused launch (e.g. n50).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "batched_problem.h"
#include "sharded_problem.h"

namespace {

// SplitMix64 stream of rocket_path_amd/problems.py (monotone distribution)
uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
double u01(uint64_t seed, uint64_t ctr) { return (double)(mix(seed + ctr) >> 11) * (1.0 / 9007199254740992.0); }

bool g_dirty = false;
void postRedisplay() { g_dirty = true; }

}  // namespace

int main(int argc, char **argv)
{
    size_t n = 1;
    uint64_t seed = 0;
    bool haveSeed = false, solve = false, f32 = false, startF4 = false;
    int gpus = 0;
    size_t watch = 0;
    unsigned redraws = 0;
    int benchPasses = 0, benchWarmup = 3;
    std::string keys = "s";
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--n") && i + 1 < argc) n = (size_t)strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) { seed = strtoull(argv[++i], nullptr, 10); haveSeed = true; }
        else if (!strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
        else if (!strcmp(argv[i], "--solve")) solve = true;
        else if (!strcmp(argv[i], "--f32")) f32 = true;
        else if (!strcmp(argv[i], "--f4")) startF4 = true;
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--watch") && i + 1 < argc) watch = (size_t)strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--bench") && i + 1 < argc) benchPasses = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) benchWarmup = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (n == 0) { fprintf(stderr, "--n must be positive\n"); return 2; }

    if (gpus > 0) {     // the sharded F3 problem: keys and the summary only (no per-lane print)
        ShardedOneDPathIP sharded(n, gpus, RP_VARIANT_F3, f32 ? RP_DTYPE_F32 : RP_DTYPE_F64);
        if (!sharded.ok()) return 1;
        sharded.init();
        if (haveSeed) {
            std::vector<double> p0(n), p1(n), p2(n);
            for (size_t i = 0; i < n; ++i) {
                p0[i] = 1000.0 * u01(seed, 3 * i + 1);
                p1[i] = p0[i] + 10.0 + 500.0 * u01(seed, 3 * i + 2);
                p2[i] = p1[i] + 10.0 + 500.0 * u01(seed, 3 * i + 3);
            }
            sharded.setProblems(p0.data(), p1.data(), p2.data());
        }
        if (benchPasses > 0) {
            if (!haveSeed) { fprintf(stderr, "--bench needs --seed (per-problem positions)\n"); return 2; }
            ShardedOneDPathIP::BenchResult br;
            if (!sharded.bench(benchPasses, benchWarmup, 1e-8, 200, br)) return 1;
            const double perGpu = (double)n / gpus;
            printf("{\"metric\": \"interior-point Newton steps/sec (whole node) + achieved HBM GB/s, 1M-problem batch\", \"value\": %.6g, "
                   "\"unit\": \"Newton steps/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6g, "
                   "\"higher_is_better\": true, \"scaling\": \"weak\", \"vs_baseline\": null, \"dtype\": \"%s\", \"data\": \"synthetic\", "
                   "\"config\": {\"workload\": \"single-process host (ShardedOneDPathIP): %.0f F3 problems per GPU, convergence-gated (gap < 1e-8, cap 200), "
                   "monotone seeded positions, feasible-start rule, one fused launch per shard per pass; each pass restarts its shard on the device\", "
                   "\"problems_total\": %zu, \"newton_steps_per_pass\": %.0f, \"converged\": %.0f, \"max_gap\": %.6g, \"max_residual_sq\": %.6g, "
                   "\"sharding\": \"contiguous shards, no data-path collective; one grouped 32-byte RCCL all-reduce (ncclCommInitAll)\"}, "
                   "\"per_device_ms_per_pass\": [",
                   br.stepsTotal / br.seconds, gpus, benchPasses, benchWarmup, br.msPerPass, f32 ? "f32" : "f64", perGpu, n,
                   br.stepsTotal / benchPasses, br.converged, br.maxGap, br.maxResidualSq);
            for (size_t d = 0; d < br.deviceMs.size(); ++d) printf("%s%.6g", d ? ", " : "", br.deviceMs[d]);
            printf("], \"devices\": [");      // which GPU every shard really sat on (rp_device_id: PCI bus id + UUID), as bench.py prints it
            for (int d = 0; d < gpus; ++d) {
                char id[128] = "?";
                if (rp_device_id(d, id, sizeof id) != RP_OK) snprintf(id, sizeof id, "device %d: %s", d, rp_last_error());
                printf("%s\"%s\"", d ? ", " : "", id);
            }
            printf("]}\n");
            return 0;
        }
        sharded.onActivate();
        if (solve) sharded.solve();
        std::istringstream in(keys);
        std::string tok;
        while (in >> tok) {
            if (tok == "SPACE") sharded.onKey(' ');
            else if (tok.size() > 1 && tok[0] == 'n') sharded.step(atoi(tok.c_str() + 1));
            else if (tok.size() == 1) sharded.onKey((unsigned char)tok[0]);
            else { fprintf(stderr, "unknown key token %s\n", tok.c_str()); return 2; }
        }
        return 0;
    }

    BatchedOneDPathIP problem3(n, RP_VARIANT_F3, f32 ? RP_DTYPE_F32 : RP_DTYPE_F64);
    BatchedOneDPathIP problem4(n, RP_VARIANT_F4, f32 ? RP_DTYPE_F32 : RP_DTYPE_F64);
    if (!problem3.ok() || !problem4.ok()) return 1;
    problem3.setRepaintHook(postRedisplay);
    problem4.setRepaintHook(postRedisplay);
    problem3.watch(watch);
    problem4.watch(watch);
    Problem *problems[] = {nullptr, nullptr, &problem3, &problem4};   // F1, F2 out of scope
    Problem *cur = startF4 ? problems[3] : problems[2];               // F3 is the reference's default (rocket_path.cpp:46)

    for (Problem *p : problems) if (p) p->init();
    if (haveSeed) {
        std::vector<double> p0(n), p1(n), p2(n);
        for (size_t i = 0; i < n; ++i) {
            p0[i] = 1000.0 * u01(seed, 3 * i + 1);
            p1[i] = p0[i] + 10.0 + 500.0 * u01(seed, 3 * i + 2);
            p2[i] = p1[i] + 10.0 + 500.0 * u01(seed, 3 * i + 3);
        }
        problem3.setProblems(p0.data(), p1.data(), p2.data());
        problem4.setProblems(p0.data(), p1.data(), p2.data());
    }
    cur->onActivate();
    if (solve) static_cast<BatchedOneDPathIP *>(cur)->solve();

    std::istringstream in(keys);
    std::string tok;
    while (in >> tok) {
        if (tok == "F3" || tok == "F4") {
            Problem *p = problems[tok == "F3" ? 2 : 3];
            if (p != cur) { cur = p; cur->onActivate(); postRedisplay(); }      // rocket_path.cpp:151-156
        } else if (tok == "p") {
            BatchedOneDPathIP *b = static_cast<BatchedOneDPathIP *>(cur);
            printf("Plot:");
            for (double x : b->plotPositions()) printf(" %.17g", x);
            printf(" |");
            for (double x : b->plotAccelerations()) printf(" %.17g", x);
            printf("\n");
        } else if (tok == "SPACE") cur->onKey(' ');
        else if (tok == "HOME") cur->onSpecialKey(RP_KEY_HOME);
        else if (tok == "END") cur->onSpecialKey(RP_KEY_END);
        else if (tok == "PGUP") cur->onSpecialKey(RP_KEY_PAGE_UP);
        else if (tok == "PGDN") cur->onSpecialKey(RP_KEY_PAGE_DOWN);
        else if (tok == "LEFT") cur->onSpecialKey(RP_KEY_LEFT);
        else if (tok == "RIGHT") cur->onSpecialKey(RP_KEY_RIGHT);
        else if (tok == "UP") cur->onSpecialKey(RP_KEY_UP);
        else if (tok == "DOWN") cur->onSpecialKey(RP_KEY_DOWN);
        else if (tok.size() > 1 && tok[0] == 'n') static_cast<BatchedOneDPathIP *>(cur)->step(atoi(tok.c_str() + 1));
        else if (tok.size() == 1) cur->onKey((unsigned char)tok[0]);
        else { fprintf(stderr, "unknown key token %s\n", tok.c_str()); return 2; }
        if (g_dirty) {      // the display callback
            g_dirty = false;
            cur->onDraw();
            ++redraws;
        }
    }
    fprintf(stderr, "redraws: %u\n", redraws);
    return 0;
}
