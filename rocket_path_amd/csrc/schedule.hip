// schedule.hip -- the scheduled order of a batch (gfx950): where each problem lies inside the field arrays.
//
// The gated solve gives every wave 64 consecutive positions of the batch (ip_kernels.hip, k_solve_chunks) and a wave
// runs until its slowest lane has converged, so the problems are kept grouped by what predicts their gated step count.
// Measured on the benchmark distribution (oracle step counts of 1,048,576 problems, tests/checks/idle_lanes.py): the count
// is a function of the segment-length ratio r = min|dX| / max|dX| (14 steps for r < 0.5 rising to 18 at r = 1) and, within a
// ratio class, of the longer segment's length.  An 11-bit key -- 64 ratio classes x 32 length levels, 4 per octave over
// [4, 1024), clamped outside -- leaves 1.5 % of the lane-steps idle (batch order: 19.2 %; 64 levels, 4,096 keys: 1.3 %; the
// 32-bit key round 2 sorted on: 1.0 %; 8,192 keys: 0.9 %): the pass is 5 us shorter with 2,048 keys than with 4,096 (half
// the histogram matrix), the solve 0.5 us longer.  A problem whose middle node lies OUTSIDE its end nodes (the path reverses; none in the
// benchmark distribution, a third of the non-monotone stress set) takes 12 steps whatever its ratio and lengths
// (174,485 of 174,696 such problems; the rest 13), two fewer than any monotone problem: all reversals share key 0.  Without
// that the stress set idles 12.6 % of its lane-steps, with it 1.3 % (profiles/r3_idle_lanes.log).
//
// With 2,048 possible keys the order is ONE stable counting sort, three small hand-written kernels on the batch's own
// stream (round 2 called rocPRIM's 32-bit radix sort here: 23 dispatches and 188 us at 1 Mi problems for an order whose
// only purpose is to save ~30 us of the solve):
//   k_sched_count    per tile of kTileProblems problems: key of every problem (kept, 2 B), histogram in LDS -> hist[tile][key];
//                    for set_problems also the problem's three positions as one 32-byte record, in PROBLEM order (coalesced):
//                    the caller's arrays are not needed after this kernel
//   k_sched_scan     per key: exclusive prefix over the tiles (in place) and the key's total
//   k_sched_scatter  per tile: base of every key (prefix of the totals) + the tile's prefix + a stable rank inside the tile
//                    -> position of every problem; writes slot_of[problem] (coalesced) and prob_of[position] (4 B scattered:
//                    the only scattered access of the pass)
// TWO FORMS OF THE THREE KERNELS (round 6), the same order from both (the stable sort by key is unique):
//   fat::   256-thread blocks, 4,096-problem tiles: the form of rounds 3-5, fastest on an otherwise idle device (32-33 us at 1 Mi
//           problems) -- what rp_batch_set_problems(_device) and rp_batch_set_state run
//   slim::  every block ONE wave, 2,048-problem tiles: 63 us alone, but able to run BESIDE a solve -- what rp_pipeline runs.  Under a
//           pipeline (the pass of batch i + 1 on one stream while batch i's solve runs on another) the fat form overlaps nothing: the solve holds all four wave slots and all 512 VGPRs of every SIMD with single-wave blocks, a slot
// that frees is refilled at once by the solve's next block, and a 256-thread block -- four slots on ONE compute unit at the
// same moment -- never fits until the solve's grid has run dry: rocprofv3 showed k_sched_scan taking 150 us (5.7 alone) and
// ending with the solve (profiles/r6_pipeline_overlap_256thread_sched_inline.log).  A one-wave block of <= 128 VGPRs fits wherever
// one solve wave has left (profiles/r6_pipeline_overlap_slim_sched_*.log: the solves of consecutive jobs then overlap).
// Both forms are checked against numpy's stable argsort (tests/test_gpu_boundary.py::test_scheduled_order_is_the_stable_sort_by_ratio_class_and_length).
// Nothing else moves: the consumer of a fresh batch -- the fused solve, or the kernel that writes the feasible start out --
// finds its problem through prob_of and reads that problem's record (one aligned 32-byte sector).  In the solve that gather
// rides under the arithmetic of the other resident waves.  Measured alternatives at 1 Mi problems
// (profiles/r3_sched_probe.log): scattering the three positions into the SoA constant fields at their position (3 x 8 B per
// problem, partial sectors) +35 us, scattering 32-byte records +17 us, the 4-byte inverse map +5 us.
// Ties keep problem order (stable at every level: tiles in order, 64-problem groups of a tile in order, lanes by a match-any
// rank), so the order is a pure function of the positions.
#include "ip_kernels.h"

#include <hip/hip_runtime.h>

namespace rp {

namespace {

#ifndef RP_SCHED_LEVEL_BITS
#define RP_SCHED_LEVEL_BITS 5                 // length levels: 5 bits = 4 per octave over [4, 1024) (6: 8 per octave, 4,096 keys -- 1.3 % idle lane-steps
                                              // instead of 1.5 %, but a pass of 36.8 instead of 31.8 us at 1 Mi problems: profiles/r3_tuning.md)
#endif
constexpr int kLevelBits = RP_SCHED_LEVEL_BITS;
constexpr int kKeyBits = 6 + kLevelBits;
constexpr int kKeys = 1 << kKeyBits;          // 64 ratio classes x 32 length levels = 2,048 keys

// key = ratio class (6 bits) : length level (kLevelBits = 5 bits).  Level = 4 per octave of the longer segment's length from 4
// upwards (exponent and top two mantissa bits of its float pattern), clamped to [0, 31].  Equal segments, zero-length pairs
// and NaN go to the last class; a reversal (segments of opposite sign) has key 0.
__device__ __forceinline__ uint32_t schedule_key(double p0, double p1, double p2)
{
    if ((p1 - p0) * (p2 - p1) < 0.0) return 0u;
    const double d0 = __builtin_fabs(p1 - p0), d1 = __builtin_fabs(p2 - p1);
    const double lo = d0 < d1 ? d0 : d1, hi = d0 < d1 ? d1 : d0;
    const double r = lo / hi * 64.0;
    const uint32_t cls = (r >= 0.0 && r < 64.0) ? (uint32_t)r : 63u;
    const float len = (float)hi;
    int lvl = 0;
    constexpr int mant = kLevelBits - 3;      // mantissa bits in a level: 2 (4 per octave)
    if (len == len && len > 0.0f) lvl = (int)(__float_as_uint(len) >> (23 - mant)) - ((127 + 2) << mant);      // 4 * (log2 floor - 2) + 2 mantissa bits
    lvl = lvl < 0 ? 0 : lvl > (1 << kLevelBits) - 1 ? (1 << kLevelBits) - 1 : lvl;
    return (cls << kLevelBits) | (uint32_t)lvl;
}

// lanes of the wave that hold the same key (kKeyBits = 11 bits) as this lane
__device__ __forceinline__ unsigned long long match_key(uint32_t key)
{
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int bit = 0; bit < kKeyBits; ++bit) {
        const bool set = (key >> bit) & 1u;
        const unsigned long long b = __ballot(set);
        peers &= set ? b : ~b;
    }
    return peers;
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// ---------------------------------------------------------------------------------------------------------------------
namespace fat {

constexpr int kThreads = 256;                 // 4 waves
constexpr int kPerThread = 16;
constexpr int kTileProblems = kThreads * kPerThread;      // 4,096 problems per tile
#ifndef RP_SCHED_COUNT_THREADS
#define RP_SCHED_COUNT_THREADS 256      // measured at 1 Mi problems: 256 threads 37.0 us per pass, 512: 37.1, 1,024: 36.3 -- not bound by its occupancy
#endif
constexpr int kCountThreads = RP_SCHED_COUNT_THREADS;      // the counting kernel's block (a tile of 4,096 problems either way)
constexpr int kCountPerThread = kTileProblems / kCountThreads;
constexpr int kWaves = kThreads / 64;
constexpr int kWaveSpan = kTileProblems / kWaves;         // consecutive problems a wave owns: 1,024 = 16 groups of 64

template <bool RECORDS>
__global__ void __launch_bounds__(kCountThreads)
k_sched_count(const double *__restrict__ pos0, const double *__restrict__ pos1, const double *__restrict__ pos2, size_t pstride,
              size_t n, uint32_t *__restrict__ hist, uint16_t *__restrict__ keys, StartRecord *__restrict__ records,
              unsigned long long *__restrict__ counters)
{
    __shared__ uint32_t s_cnt[kKeys];
    for (int k = threadIdx.x; k < kKeys; k += kCountThreads) s_cnt[k] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 128 && counters) counters[threadIdx.x] = 0;      // the batch's progress counters: a new problem set
    __syncthreads();
    const size_t first = (size_t)blockIdx.x * kTileProblems;
    double p0[kCountPerThread], p1[kCountPerThread], p2[kCountPerThread];
#pragma unroll
    for (int q = 0; q < kCountPerThread; ++q) {
        const size_t i = first + (size_t)q * kCountThreads + threadIdx.x;      // any assignment of problems to threads: only the counts matter
        const size_t at = (i < n ? i : 0) * pstride;
        p0[q] = pos0[at]; p1[q] = pos1[at]; p2[q] = pos2[at];
    }
#pragma unroll
    for (int q = 0; q < kCountPerThread; ++q) {
        const size_t i = first + (size_t)q * kCountThreads + threadIdx.x;
        if (i < n) {
            const uint32_t key = schedule_key(p0[q], p1[q], p2[q]);
            atomicAdd(&s_cnt[key], 1u);
            keys[i] = (uint16_t)key;
            if constexpr (RECORDS) {
                typedef double v2 __attribute__((ext_vector_type(2)));
                v2 *rec = reinterpret_cast<v2 *>(records + i);
                const v2 a = {p0[q], p1[q]}, b = {p2[q], __longlong_as_double((long long)i)};
                __builtin_nontemporal_store(a, rec);
                __builtin_nontemporal_store(b, rec + 1);
            }
        }
    }
    __syncthreads();
    uint32_t *row = hist + (size_t)blockIdx.x * kKeys;
    for (int k = threadIdx.x; k < kKeys; k += kCountThreads) row[k] = s_cnt[k];
}

// hist[tile][key] -> exclusive prefix over the tiles, per key, in place; total[key].  A block owns 16 keys (one 64-byte
// segment of every row); its 256 threads are 16 keys x 16 groups of consecutive rows.
__global__ void __launch_bounds__(kThreads)
k_sched_scan(uint32_t *__restrict__ hist, unsigned ntiles, uint32_t *__restrict__ total)
{
    __shared__ uint32_t s_seg[16][17];
    const unsigned key = blockIdx.x * 16 + (threadIdx.x & 15);
    const unsigned g = threadIdx.x >> 4;
    const unsigned per = (ntiles + 15) / 16;
    const unsigned r0 = g * per, r1 = (r0 + per < ntiles) ? r0 + per : ntiles;
    constexpr unsigned kInRegs = 16;      // up to 256 tiles (1 Mi problems) a thread's rows stay in registers between the two sweeps
    uint32_t held[kInRegs];
    uint32_t sum = 0;
    if (per <= kInRegs) {
#pragma unroll
        for (unsigned q = 0; q < kInRegs; ++q) {
            held[q] = (r0 + q < r1) ? hist[(size_t)(r0 + q) * kKeys + key] : 0u;
            sum += held[q];
        }
    } else {
        for (unsigned r = r0; r < r1; ++r) sum += hist[(size_t)r * kKeys + key];
    }
    s_seg[g][threadIdx.x & 15] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (unsigned q = 0; q < g; ++q) run += s_seg[q][threadIdx.x & 15];
    if (g == 15) total[key] = run + sum;
    if (per <= kInRegs) {
#pragma unroll
        for (unsigned q = 0; q < kInRegs; ++q) {
            if (r0 + q < r1) hist[(size_t)(r0 + q) * kKeys + key] = run;
            run += held[q];
        }
    } else {
        for (unsigned r = r0; r < r1; ++r) {
            const size_t at = (size_t)r * kKeys + key;
            const uint32_t v = hist[at];
            hist[at] = run;
            run += v;
        }
    }
}

__global__ void __launch_bounds__(kThreads)
k_sched_scatter(const uint16_t *__restrict__ keys, size_t n, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ total,
                uint32_t *__restrict__ slot_of, uint32_t *__restrict__ prob_of)
{
    __shared__ uint32_t s_off[kKeys];                 // position of the tile's first problem with this key
    __shared__ uint32_t s_wave[kWaves][kKeys / 2];    // per wave and key: problems of earlier waves and groups of the tile (two 16-bit counts per word)
    __shared__ uint32_t s_part[kThreads];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int k = tid; k < kWaves * (kKeys / 2); k += kThreads) (&s_wave[0][0])[k] = 0;
    __syncthreads();

    // this wave's problems, in order: group q = problems [first + w * 1024 + 64 q, + 64)
    const size_t first = (size_t)blockIdx.x * kTileProblems + (size_t)w * kWaveSpan;
    uint32_t key[kPerThread];
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const size_t i = first + (size_t)q * 64 + lane;
        key[q] = i < n ? (uint32_t)keys[i] : (uint32_t)(kKeys - 1);
    }
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const size_t i = first + (size_t)q * 64 + lane;
        if (i < n) atomicAdd(&s_wave[w][key[q] >> 1], 1u << (16 * (key[q] & 1u)));      // <= 1,024 per wave and key: fits 16 bits
    }

    // base of every key = exclusive prefix of the totals (each tile recomputes it: 16 KiB from L2) + this tile's prefix
    {
        constexpr int kMine = kKeys / kThreads;      // keys per thread: 16
        uint32_t t[kMine], sum = 0;
#pragma unroll
        for (int j = 0; j < kMine; ++j) { t[j] = total[tid * kMine + j]; sum += t[j]; }
        // exclusive prefix of the 256 partial sums: inside the wave by shuffles, across the four waves through LDS
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_part[w] = incl;
        __syncthreads();                               // also: every wave's counts are in s_wave
        uint32_t run = incl - sum;
        for (int j = 0; j < w; ++j) run += s_part[j];
        const uint32_t *row = hist + (size_t)blockIdx.x * kKeys + tid * kMine;
#pragma unroll
        for (int j = 0; j < kMine; ++j) {
            const int k = tid * kMine + j;
            s_off[k] = run + row[j];
            run += t[j];
            // per-wave counts -> exclusive prefix over the waves (in place, both halves of the word at once)
            if ((k & 1) == 0) {
                uint32_t acc = 0;
#pragma unroll
                for (int v = 0; v < kWaves; ++v) { const uint32_t c = s_wave[v][k >> 1]; s_wave[v][k >> 1] = acc; acc += c; }
            }
        }
    }
    __syncthreads();

    // stable rank, group by group: lanes with the same key are ranked by lane, the first of them moves the wave's counter
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const size_t i = first + (size_t)q * 64 + lane;
        const bool live = i < n;
        const uint32_t k = live ? key[q] : (uint32_t)(kKeys - 1);      // dead lanes only ever sit behind live ones (the tail of the batch)
        unsigned long long peers = match_key(k) & __ballot(live);
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        const int rank = __popcll(below);
        uint32_t before = 0;
        if (live && rank == 0) before = atomicAdd(&s_wave[w][k >> 1], (uint32_t)__popcll(peers) << (16 * (k & 1u)));
        const int leader = peers ? __ffsll((long long)peers) - 1 : 0;
        before = __shfl(before, leader);
        before = (before >> (16 * (k & 1u))) & 0xffffu;
        if (live) {
            const uint32_t at = s_off[k] + before + (uint32_t)rank;
            slot_of[i] = at;
            prob_of[at] = (uint32_t)i;
        }
    }
}

inline unsigned tiles_for(size_t n) { return (unsigned)((n + kTileProblems - 1) / kTileProblems); }

}  // namespace fat

// ---------------------------------------------------------------------------------------------------------------------
namespace slim {

constexpr int kThreads = 64;                  // ONE wave per block, in all three kernels (see the top of this file)
#ifndef RP_SCHED_TILE
#define RP_SCHED_TILE 2048                    // problems per tile: 512 tiles at 1 Mi problems (a 4 MB histogram matrix)
#endif
constexpr int kTileProblems = RP_SCHED_TILE;
constexpr int kGroups = kTileProblems / 64;   // 64-problem groups of a tile, walked in order by its one wave
constexpr int kChunk = 16;                    // groups whose key loads are issued together (k_sched_scatter)
constexpr int kCountChunk = 8;                // ... and whose position loads are (k_sched_count: 8 x 3 doubles in flight per lane; 16 made it 146 VGPRs --
                                              // more than the 128 one retiring solve wave leaves behind, and then the block cannot slip in beside a running solve)
static_assert(kTileProblems % (64 * kChunk) == 0 && kTileProblems <= 65535, "a tile is whole chunks of groups, and a key's count in a tile fits 16 bits");
constexpr int kKeysPerLane = kKeys / 64;      // 32

template <bool RECORDS>
__global__ void __launch_bounds__(kThreads, 4)      // <= 128 VGPRs: fits the registers of ONE solve wave
k_sched_count(const double *__restrict__ pos0, const double *__restrict__ pos1, const double *__restrict__ pos2, size_t pstride,
              size_t n, uint32_t *__restrict__ hist, uint16_t *__restrict__ keys, StartRecord *__restrict__ records,
              unsigned long long *__restrict__ counters)
{
    __shared__ uint32_t s_cnt[kKeys];
    const int lane = threadIdx.x;
#pragma unroll
    for (int j = 0; j < kKeysPerLane; ++j) s_cnt[j * 64 + lane] = 0;
    if (blockIdx.x == 0 && counters) { counters[lane] = 0; counters[64 + lane] = 0; }      // the batch's progress counters: a new problem set
    __syncthreads();
    const size_t first = (size_t)blockIdx.x * kTileProblems;
    for (int c = 0; c < kGroups; c += kCountChunk) {
        double p0[kCountChunk], p1[kCountChunk], p2[kCountChunk];
#pragma unroll
        for (int q = 0; q < kCountChunk; ++q) {
            const size_t i = first + (size_t)(c + q) * 64 + lane;
            const size_t at = (i < n ? i : 0) * pstride;
            p0[q] = pos0[at]; p1[q] = pos1[at]; p2[q] = pos2[at];
        }
#pragma unroll
        for (int q = 0; q < kCountChunk; ++q) {
            const size_t i = first + (size_t)(c + q) * 64 + lane;
            if (i < n) {
                const uint32_t key = schedule_key(p0[q], p1[q], p2[q]);
                atomicAdd(&s_cnt[key], 1u);
                keys[i] = (uint16_t)key;
                if constexpr (RECORDS) {
                    typedef double v2 __attribute__((ext_vector_type(2)));
                    v2 *rec = reinterpret_cast<v2 *>(records + i);
                    const v2 a = {p0[q], p1[q]}, b = {p2[q], __longlong_as_double((long long)i)};
                    __builtin_nontemporal_store(a, rec);
                    __builtin_nontemporal_store(b, rec + 1);
                }
            }
        }
    }
    __syncthreads();
    uint32_t *row = hist + (size_t)blockIdx.x * kKeys;
#pragma unroll
    for (int j = 0; j < kKeysPerLane; ++j) row[j * 64 + lane] = s_cnt[j * 64 + lane];
}

// hist[tile][key] -> exclusive prefix over the tiles, per key, in place; total[key].  A block owns kScanKeys = 4 keys (16 bytes of
// every row) and its 64 threads are 4 keys x 16 groups of consecutive rows, each walking its rows twice (sum, then prefix: the
// matrix is a few MB and sits in L2).  512 one-wave blocks of 2 x 32 dependent round trips each: beside a running solve the
// pass is a chain of latencies, and this is the link that was longest (128 blocks of 2 x 128: 62 us against count's 55).
constexpr int kScanKeys = 4, kScanGroups = 64 / kScanKeys;
__global__ void __launch_bounds__(kThreads)
k_sched_scan(uint32_t *__restrict__ hist, unsigned ntiles, uint32_t *__restrict__ total)
{
    __shared__ uint32_t s_seg[kScanGroups][kScanKeys];
    const unsigned kq = threadIdx.x & (kScanKeys - 1), key = blockIdx.x * kScanKeys + kq;
    const unsigned g = threadIdx.x / kScanKeys;
    const unsigned per = (ntiles + kScanGroups - 1) / kScanGroups;
    const unsigned r0 = g * per < ntiles ? g * per : ntiles, r1 = (r0 + per < ntiles) ? r0 + per : ntiles;
    uint32_t sum = 0;
    unsigned r = r0;
    for (; r + 8 <= r1; r += 8) {      // eight rows' loads in flight
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = hist[(size_t)(r + q) * kKeys + key];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += v[q];
    }
    for (; r < r1; ++r) sum += hist[(size_t)r * kKeys + key];
    s_seg[g][kq] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (unsigned q = 0; q < g; ++q) run += s_seg[q][kq];
    if (g == kScanGroups - 1) total[key] = run + sum;
    r = r0;
    for (; r + 8 <= r1; r += 8) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = hist[(size_t)(r + q) * kKeys + key];
#pragma unroll
        for (int q = 0; q < 8; ++q) { hist[(size_t)(r + q) * kKeys + key] = run; run += v[q]; }
    }
    for (; r < r1; ++r) {
        const size_t at = (size_t)r * kKeys + key;
        const uint32_t v = hist[at];
        hist[at] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(kThreads, 4)
k_sched_scatter(const uint16_t *__restrict__ keys, size_t n, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ total,
                uint32_t *__restrict__ slot_of, uint32_t *__restrict__ prob_of)
{
    __shared__ uint32_t s_off[kKeys];                 // position of the tile's next problem with this key
    const int lane = threadIdx.x;
    // base of every key = exclusive prefix of the totals (each tile recomputes it: 8 KiB from L2) + this tile's prefix.  Keys are
    // dealt to the lanes 64 apart (coalesced loads); the prefix runs over key order: 32 rounds of a 64-lane scan.
    {
        const uint32_t *row = hist + (size_t)blockIdx.x * kKeys;
        uint32_t t[kKeysPerLane], h[kKeysPerLane];
#pragma unroll
        for (int j = 0; j < kKeysPerLane; ++j) { t[j] = total[j * 64 + lane]; h[j] = row[j * 64 + lane]; }
        uint32_t run = 0;                             // keys below this round's 64
#pragma unroll
        for (int j = 0; j < kKeysPerLane; ++j) {
            uint32_t incl = t[j];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            s_off[j * 64 + lane] = run + incl - t[j] + h[j];
            run += __shfl(incl, 63);
        }
    }
    __syncthreads();

    // stable rank, group by group: lanes with the same key are ranked by lane, the first of them moves the key's counter (an LDS
    // atomic: groups follow each other in the one wave, and an atomic's result is what the previous group's left)
    const size_t first = (size_t)blockIdx.x * kTileProblems;
    for (int c = 0; c < kGroups; c += kChunk) {
        uint32_t key[kChunk];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const size_t i = first + (size_t)(c + q) * 64 + lane;
            key[q] = i < n ? (uint32_t)keys[i] : (uint32_t)(kKeys - 1);
        }
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const size_t i = first + (size_t)(c + q) * 64 + lane;
            const bool live = i < n;
            const uint32_t k = live ? key[q] : (uint32_t)(kKeys - 1);      // dead lanes only ever sit behind live ones (the tail of the batch)
            const unsigned long long peers = match_key(k) & __ballot(live);
            const unsigned long long below = peers & ((1ull << lane) - 1ull);
            const int rank = __popcll(below);
            uint32_t at = 0;
            if (live && rank == 0) at = atomicAdd(&s_off[k], (uint32_t)__popcll(peers));
            const int leader = peers ? __ffsll((long long)peers) - 1 : 0;
            at = __shfl(at, leader) + (uint32_t)rank;
            if (live) {
                slot_of[i] = at;
                prob_of[at] = (uint32_t)i;
            }
        }
    }
}

inline unsigned tiles_for(size_t n) { return (unsigned)((n + kTileProblems - 1) / kTileProblems); }

}  // namespace slim

}  // namespace

// scratch layout: hist[tiles][kKeys] | total[kKeys] | keys[n] (16 bit); sized for the form with the smaller tiles
hipError_t schedule_scratch_bytes(size_t n, size_t *bytes)
{
    const unsigned tiles = slim::tiles_for(n) > fat::tiles_for(n) ? slim::tiles_for(n) : fat::tiles_for(n);
    *bytes = align256((size_t)tiles * kKeys * sizeof(uint32_t)) + kKeys * sizeof(uint32_t) + align256(n * sizeof(uint16_t));
    return hipSuccess;
}

hipError_t launch_schedule(const BatchView &b, const double *d_pos0, const double *d_pos1, const double *d_pos2, size_t pstride,
                           bool write_records, void *d_scratch, size_t scratch_bytes, hipStream_t stream, bool one_wave_blocks)
{
    if (b.n == 0) return hipSuccess;
    size_t need = 0;
    (void)schedule_scratch_bytes(b.n, &need);
    if (scratch_bytes < need || (write_records && !b.records)) return hipErrorInvalidValue;
    const unsigned tiles = one_wave_blocks ? slim::tiles_for(b.n) : fat::tiles_for(b.n);
    uint32_t *hist = (uint32_t *)d_scratch;
    uint32_t *total = (uint32_t *)((char *)d_scratch + align256((size_t)tiles * kKeys * sizeof(uint32_t)));
    uint16_t *keys = (uint16_t *)(total + kKeys);
    StartRecord *rec = write_records ? b.records : (StartRecord *)nullptr;
    if (one_wave_blocks) {
        if (write_records) hipLaunchKernelGGL((slim::k_sched_count<true>), dim3(tiles), dim3(slim::kThreads), 0, stream, d_pos0, d_pos1, d_pos2, pstride, b.n, hist, keys, rec, b.counters);
        else               hipLaunchKernelGGL((slim::k_sched_count<false>), dim3(tiles), dim3(slim::kThreads), 0, stream, d_pos0, d_pos1, d_pos2, pstride, b.n, hist, keys, rec, b.counters);
        hipLaunchKernelGGL(slim::k_sched_scan, dim3(kKeys / slim::kScanKeys), dim3(slim::kThreads), 0, stream, hist, tiles, total);
        hipLaunchKernelGGL(slim::k_sched_scatter, dim3(tiles), dim3(slim::kThreads), 0, stream, (const uint16_t *)keys, b.n, (const uint32_t *)hist,
                           (const uint32_t *)total, b.slot_of, b.prob_of);
    } else {
        if (write_records) hipLaunchKernelGGL((fat::k_sched_count<true>), dim3(tiles), dim3(fat::kCountThreads), 0, stream, d_pos0, d_pos1, d_pos2, pstride, b.n, hist, keys, rec, b.counters);
        else               hipLaunchKernelGGL((fat::k_sched_count<false>), dim3(tiles), dim3(fat::kCountThreads), 0, stream, d_pos0, d_pos1, d_pos2, pstride, b.n, hist, keys, rec, b.counters);
        hipLaunchKernelGGL(fat::k_sched_scan, dim3(kKeys / 16), dim3(fat::kThreads), 0, stream, hist, tiles, total);
        hipLaunchKernelGGL(fat::k_sched_scatter, dim3(tiles), dim3(fat::kThreads), 0, stream, (const uint16_t *)keys, b.n, (const uint32_t *)hist,
                           (const uint32_t *)total, b.slot_of, b.prob_of);
    }
    return hipGetLastError();
}

}  // namespace rp
