// feas_core.h -- per-lane device code of the feasibility move (the Space key).
//
// Reference: moveTowardFeasibility, onedpath_ip.cpp:648-721 (F4: onedpath2_ip.cpp:536-609):
// collect the violated constraints (error > 0), form the Gram matrix a = G G^T of their
// gradients, m = a.colPivHouseholderQr().solve(err), dX = G^T (-m), var[0..2] += dX.
//
// Unlike the Newton step this is NOT arithmetic-bound and NOT well conditioned: the Gram
// matrix squares the conditioning of G, and with four violated rows (three variables) it is
// singular up to rounding, so whether its last pivot counts (ColPivHouseholderQR.h:524-525)
// is decided by rounding noise.  Only the reference's own sequence of roundings reproduces
// the reference there.  So this file is an operation-for-operation transcription: IEEE
// divisions and square roots, no fused multiply-adds (the build has -ffp-contract=off and
// nothing below calls fma_), the accelerations in the reference's expression order
// (onedpath_ip.cpp:383-391, 424-432), and the rank-revealing column-pivoted Householder QR of
// ColPivHouseholderQR.h:480-611 / Householder.h:65-131 step by step with every reduction in the order
// Eigen's DYNAMIC-size kernels use (the reference works on MatrixXd / VectorXd here):
//   * squaredNorm and the 1 x 1 inner product: Redux.h:209-263, linear-vectorised traversal, SSE2 packets of
//     two, split independent of the address (esqn_ below);
//   * essential^T * bottom and g^T * m go through the row-major matrix-vector kernel
//     (GeneralMatrixVector.h:364-612), which sums scalars up to the vector operand's first 16-byte aligned
//     element, then packet accumulators, then the scalar tail.  Eigen's buffers are 16-byte aligned, so that
//     split is a function of the operand's linear offset n k + k + 1 in the n x n matrix: for n <= 4 everything
//     comes out sequential except (n = 4, k = 0), x1 + (x2 + x3), and the final 4-term g^T m,
//     (x0 + x2) + (x1 + x3).
// The same sequence is what oracle/ip_oracle.c restates (orc_colpiv_qr_solve_dynamic), which in turn is pinned bit
// for bit on the reference's own moveTowardFeasibility compiled in the build container (oracle/_ref), four violated
// rows (rank-deficient Gram matrix) included; the GPU tests compare against the oracle bit for bit.
//
// At most 4 constraints can be violated at once in either variant (F3's come in -a-L / a-L
// pairs that exclude each other; F4 has 4), so the Gram matrix is at most 4x4 and everything
// runs in registers: loops are fully unrolled over NMAX = 4 with `< n` guards, and the runtime
// pivot column is applied through compare-and-select swaps, never through dynamic register
// indexing.
#pragma once

#include "ip_core.h"

namespace rp {

template <typename T> __device__ __forceinline__ T eps_();
template <> __device__ __forceinline__ double eps_<double>() { return 2.220446049250313e-16; }
template <> __device__ __forceinline__ float eps_<float>() { return 1.1920929e-07f; }
template <typename T> __device__ __forceinline__ T tiny_();
template <> __device__ __forceinline__ double tiny_<double>() { return 2.2250738585072014e-308; }
template <> __device__ __forceinline__ float tiny_<float>() { return 1.17549435e-38f; }

// squaredNorm() of e[0..m) in Eigen 3.3.0's SSE2 order (packets of two doubles; the dynamic-size traversal of
// Redux.h:209-263 and the fixed-size unroller agree up to m = 4):
// m = 1: s0;  m = 2: s0 + s1;  m = 3: (s0 + s1) + s2;  m = 4: (s0 + s2) + (s1 + s3).
template <typename T>
__device__ __forceinline__ T esqn_(T e0, T e1, T e2, T e3, int m)
{
    const T s0 = e0 * e0, s1 = e1 * e1, s2 = e2 * e2, s3 = e3 * e3;
    const T r2 = s0 + s1;
    const T r3 = r2 + s2;
    const T r4 = (s0 + s2) + (s1 + s3);
    return m <= 0 ? T(0) : m == 1 ? s0 : m == 2 ? r2 : m == 3 ? r3 : r4;
}

// A[r][c], or 0 for a row past the 4 x 4 register block (r is a compile-time constant wherever this is called)
template <typename T>
__device__ __forceinline__ T below_(const T (&A)[4][4], int r, int c) { return r < 4 ? A[r < 4 ? r : 0][c] : T(0); }

// x = colPivHouseholderQr(A).solve(b) for the leading n x n block, n <= 4.  A[r][c].
template <typename T>
__device__ __forceinline__ void colpiv_qr_solve4(int n, T (&A)[4][4], T (&b)[4], T (&x)[4])
{
    constexpr int N = 4;
    T nu[N], nd[N], tau[N];      // updated / direct column norms, Householder coefficients
    int perm[N];
    // ColPivHouseholderQR.h:502-507
#pragma unroll
    for (int c = 0; c < N; ++c) {
        nd[c] = nu[c] = sqrt_(esqn_<T>(A[0][c], A[1][c], A[2][c], A[3][c], n));
        perm[c] = c;
        tau[c] = T(0);
    }
    T maxn = nu[0];
#pragma unroll
    for (int c = 1; c < N; ++c) if (c < n && nu[c] > maxn) maxn = nu[c];
    const T thr_helper = (maxn * eps_<T>()) * (maxn * eps_<T>()) / (T)n;      // :509
    const T down_thr = sqrt_(eps_<T>());
    int nz = n;                                                               // :512

#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < n) {
            // pivot column: largest updated norm among k..n-1, first wins (:517-519)
            int big = k;
            T bigv = nu[k];
#pragma unroll
            for (int j = k + 1; j < N; ++j) if (j < n && nu[j] > bigv) { bigv = nu[j]; big = j; }
            if (nz == n && bigv * bigv < thr_helper * (T)(n - k)) nz = k;      // :524-525
#pragma unroll
            for (int j = k + 1; j < N; ++j) {                                 // :528-533
                const bool sw = (big == j);
#pragma unroll
                for (int r = 0; r < N; ++r) swap_if(sw, A[r][k], A[r][j]);
                swap_if(sw, nu[k], nu[j]);
                swap_if(sw, nd[k], nd[j]);
                const int pk = sw ? perm[j] : perm[k], pj = sw ? perm[k] : perm[j];
                perm[k] = pk;      // composing the transpositions as they happen = :574-576
                perm[j] = pj;
            }
            // makeHouseholderInPlace on rows k..n-1 of column k (Householder.h:65-94)
            const int m = n - k - 1;      // tail length
            const T tail = esqn_<T>(below_(A, k + 1, k), below_(A, k + 2, k), below_(A, k + 3, k), T(0), m);
            const T c0 = A[k][k];
            T beta, tk;
            if (tail <= tiny_<T>()) {
                tk = T(0);
                beta = c0;
#pragma unroll
                for (int r = k + 1; r < N; ++r) if (r < n) A[r][k] = T(0);
            } else {
                beta = sqrt_(c0 * c0 + tail);
                if (c0 >= T(0)) beta = -beta;
                const T den = c0 - beta;
#pragma unroll
                for (int r = k + 1; r < N; ++r) if (r < n) A[r][k] = A[r][k] / den;
                tk = (beta - c0) / beta;
            }
            tau[k] = tk;
            A[k][k] = beta;
            // applyHouseholderOnTheLeft to rows k.., columns k+1.. (Householder.h:113-131)
            if (m > 0 && tk != T(0)) {
                T tmp[N];
#pragma unroll
                for (int j = k + 1; j < N; ++j) {
                    T t = T(0);
                    if (k == 0 && n == 4) {      // the essential part starts at the odd offset 1: scalar head, then one packet
                        t = (T(0) + A[1][0] * A[1][j]) + ((T(0) + A[2][0] * A[2][j]) + (T(0) + A[3][0] * A[3][j]));      // accumulators start at 0, as the kernel's
                    } else {
#pragma unroll
                        for (int r = k + 1; r < N; ++r) if (r < n) t = t + A[r][k] * A[r][j];
                    }
                    tmp[j] = t + A[k][j];
                }
#pragma unroll
                for (int j = k + 1; j < N; ++j) if (j < n) A[k][j] = A[k][j] - tk * tmp[j];
#pragma unroll
                for (int j = k + 1; j < N; ++j)
#pragma unroll
                    for (int r = k + 1; r < N; ++r) if (j < n && r < n) A[r][j] = A[r][j] - tk * A[r][k] * tmp[j];
            }
            // column-norm downdate (ColPivHouseholderQR.h:551-571)
#pragma unroll
            for (int j = k + 1; j < N; ++j) {
                if (j < n && nu[j] != T(0)) {
                    T temp = abs_(A[k][j]) / nu[j];
                    temp = (T(1) + temp) * (T(1) - temp);
                    temp = temp < T(0) ? T(0) : temp;
                    const T temp2 = temp * ((nu[j] / nd[j]) * (nu[j] / nd[j]));
                    if (temp2 <= down_thr) {
                        nd[j] = (m > 0) ? sqrt_(esqn_<T>(below_(A, k + 1, j), below_(A, k + 2, j), below_(A, k + 3, j), T(0), m)) : T(0);
                        nu[j] = nd[j];
                    } else {
                        nu[j] = nu[j] * sqrt_(temp);
                    }
                }
            }
        }
    }
    // _solve_impl (:585-611): c = Q^T b through H_0 .. H_{nz-1}, back substitution on the leading nz x nz block,
    // zeros for the rest, un-permute
    T c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = b[i];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < nz) {
            if (n - k == 1) {
                c[k] = c[k] * (T(1) - tau[k]);
            } else if (tau[k] != T(0)) {
                T t = T(0);
#pragma unroll
                for (int r = k + 1; r < N; ++r) if (r < n) t = t + A[r][k] * c[r];
                t = t + c[k];
                c[k] = c[k] - tau[k] * t;
#pragma unroll
                for (int r = k + 1; r < N; ++r) if (r < n) c[r] = c[r] - tau[k] * A[r][k] * t;
            }
        }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        if (i < nz) {
            c[i] = c[i] / A[i][i];
#pragma unroll
            for (int j = 0; j < i; ++j) c[j] = c[j] - c[i] * A[j][i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int d = 0; d < N; ++d) if (i < nz && perm[i] == d) x[d] = c[i];
    }
}

// evalAccelInit / evalAccelFinal in the reference's expression order (onedpath_ip.cpp:383-391, 424-432): one
// segment (x0, v0) -> (x1, v1) over t.  end = 0: initial, 1: final.
template <typename T>
__device__ __forceinline__ void ref_accel(int end, T dX, T v0, T v1, T t, T &a, T &dAdT, T &dAdV0, T &dAdV1)
{
    if (end == 0) {
        a = (dX * T(6) / t + v0 * T(-4) + v1 * T(-2)) / t;
        dAdT = (dX * T(-12) / t + v0 * T(4) + v1 * T(2)) / (t * t);
        dAdV0 = T(-4) / t;
        dAdV1 = T(-2) / t;
    } else {
        a = (dX * T(-6) / t + v0 * T(2) + v1 * T(4)) / t;
        dAdT = (dX * T(12) / t + v0 * T(-2) + v1 * T(-4)) / (t * t);
        dAdV0 = T(2) / t;
        dAdV1 = T(4) / t;
    }
}

// The move for one problem: positions' differences dx0 = pos1 - pos0, dx1 = pos2 - pos1, end velocities vel0 / vel2,
// the point (vel1, t0, t1).  Returns false when nothing is violated (the reference then adds a zero dX).
template <typename T, int VARIANT>
__device__ __forceinline__ bool feasibility_move(T dX0, T dX1, T vel0, T vel2, T vel1, T t0, T t1, T L, T &dxv, T &dx0, T &dx1)
{
    constexpr int NC = CMap<VARIANT>::NC;
    // compact the violated rows into at most 4 slots, in constraint order (onedpath_ip.cpp:652-672)
    T g[4][3], er[4];
    int n = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) { g[s][0] = g[s][1] = g[s][2] = T(0); er[s] = T(0); }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int seg = c_segment<VARIANT>(i);
        const int end = VARIANT == 3 ? ((i >> 1) & 1) : (i & 1);
        T a, dAdT, dAdV0, dAdV1;
        if (seg == 0) ref_accel<T>(end, dX0, vel0, vel1, t0, a, dAdT, dAdV0, dAdV1);
        else          ref_accel<T>(end, dX1, vel1, vel2, t1, a, dAdT, dAdV0, dAdV1);
        const T dAdV = seg == 0 ? dAdV1 : dAdV0;      // vel1 is the segment's v1 (seg 0) or v0 (seg 1)
        T ci, gv, gt;
        if constexpr (VARIANT == 3) {
            if (i & 1) { ci = a - L; gv = dAdV; gt = dAdT; }
            else       { ci = -a - L; gv = -dAdV; gt = -dAdT; }
        } else {
            ci = (a * a - L * L) / T(2);
            gv = a * dAdV;
            gt = a * dAdT;
        }
        if (ci > T(0)) {
            const T g0 = gv, g1 = (seg == 0) ? gt : T(0), g2 = (seg == 0) ? T(0) : gt;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (n == s) { g[s][0] = g0; g[s][1] = g1; g[s][2] = g2; er[s] = ci; }
            ++n;
        }
    }
    dxv = dx0 = dx1 = T(0);
    if (n == 0) return false;
    if (n > 4) n = 4;   // unreachable: see header comment
    T A[4][4], m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            A[r][c] = ((T(0) + g[r][0] * g[c][0]) + g[r][1] * g[c][1]) + g[r][2] * g[c][2];
    colpiv_qr_solve4<T>(n, A, er, m);
    // dX = g^T * -m (onedpath_ip.cpp:696): the matrix-vector kernel on aligned operands, alpha = -1
    if (n == 4) {      // two packets: lanes (x0 + x2, x1 + x3)
        dxv = -((g[0][0] * m[0] + g[2][0] * m[2]) + (g[1][0] * m[1] + g[3][0] * m[3]));
        dx0 = -((g[0][1] * m[0] + g[2][1] * m[2]) + (g[1][1] * m[1] + g[3][1] * m[3]));
        dx1 = -((g[0][2] * m[0] + g[2][2] * m[2]) + (g[1][2] * m[1] + g[3][2] * m[3]));
    } else {           // n <= 3: one packet and a scalar tail = the sequential sum
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < n) {
                dxv = dxv + g[s][0] * -m[s];
                dx0 = dx0 + g[s][1] * -m[s];
                dx1 = dx1 + g[s][2] * -m[s];
            }
        }
    }
    return true;
}

}  // namespace rp
