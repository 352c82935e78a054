// schedule.hip -- the scheduled order of a batch (gfx950): where each problem lies inside the field arrays.
//
// The gated solve gives every wave 64 consecutive positions of the batch (ip_kernels.hip, k_solve_chunks) and a wave
// runs until its slowest lane has converged, so the problems are kept sorted by what predicts their gated step count.
// Measured on the benchmark distribution (oracle step counts of 1,048,576 problems): the count is a function of the
// segment-length ratio r = min|dX| / max|dX| (14 steps for r < 0.5 rising to 18 at r = 1) and, within a ratio class, of
// the longer segment's length; sorted by (class of r, length) the step counts inside a 64-problem chunk differ by 0.34 on
// average and 1.0 % of the lane-steps are idle (batch order: 19.2 %; ratio alone: 3.2 %).
//
// This runs when positions are set (set_problems / set_state), never per solve: one key per problem, one stable radix
// sort of (key, problem index) pairs -- rocPRIM through hipCUB, a library sort being the right tool for a one-off
// setup step -- and the inverse map.  Ties keep problem order, so the order is a pure function of the positions.
#include "ip_kernels.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace rp {

namespace {

constexpr int kBlock = 256;

// key = ratio class (6 bits) : the longer segment's length as the top 26 bits of its float pattern (monotone for
// positive floats).  Equal segments, zero-length pairs and NaN go to the last class.
__global__ void __launch_bounds__(kBlock)
k_schedule_keys(const double *__restrict__ pos0, const double *__restrict__ pos1, const double *__restrict__ pos2, size_t pstride,
                size_t n, uint32_t *__restrict__ keys, uint32_t *__restrict__ index)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const size_t at = i * pstride;
    const double d0 = __builtin_fabs(pos1[at] - pos0[at]), d1 = __builtin_fabs(pos2[at] - pos1[at]);
    const double lo = d0 < d1 ? d0 : d1, hi = d0 < d1 ? d1 : d0;
    const double r = lo / hi * 64.0;
    const uint32_t cls = (r >= 0.0 && r < 64.0) ? (uint32_t)r : 63u;
    const float len = (float)hi;
    const uint32_t bits = (len == len && len > 0.0f) ? (__float_as_uint(len) >> 5) : 0u;      // < 2^26 (inf: 0x3FC0000)
    keys[i] = (cls << 26) | bits;
    index[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(kBlock)
k_invert(const uint32_t *__restrict__ prob_of, size_t n, uint32_t *__restrict__ slot_of)
{
    const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (s < n) slot_of[prob_of[s]] = (uint32_t)s;
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// scratch layout: keys in | keys out | index in | the sort's temporary storage
hipError_t schedule_scratch_bytes(size_t n, size_t *bytes)
{
    size_t temp = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                      (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, 32, (hipStream_t)0);
    *bytes = 3 * align256(n * sizeof(uint32_t)) + align256(temp);
    return e;
}

hipError_t launch_schedule(const BatchView &b, const double *d_pos0, const double *d_pos1, const double *d_pos2, size_t pstride,
                           void *d_scratch, size_t scratch_bytes, hipStream_t stream)
{
    if (b.n == 0) return hipSuccess;
    const size_t words = align256(b.n * sizeof(uint32_t));
    if (scratch_bytes < 3 * words) return hipErrorInvalidValue;
    uint32_t *keys_in = (uint32_t *)d_scratch;
    uint32_t *keys_out = (uint32_t *)((char *)d_scratch + words);
    uint32_t *index_in = (uint32_t *)((char *)d_scratch + 2 * words);
    void *temp = (char *)d_scratch + 3 * words;
    size_t temp_bytes = scratch_bytes - 3 * words;
    const unsigned grid = (unsigned)((b.n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_schedule_keys, dim3(grid), dim3(kBlock), 0, stream, d_pos0, d_pos1, d_pos2, pstride, b.n, keys_in, index_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint32_t *)keys_in, keys_out, (const uint32_t *)index_in,
                                                      b.prob_of, (int)b.n, 0, 32, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_invert, dim3(grid), dim3(kBlock), 0, stream, (const uint32_t *)b.prob_of, b.n, b.slot_of);
    return hipGetLastError();
}

}  // namespace rp
