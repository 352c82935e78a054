// feas_core.h -- per-lane device code of the feasibility move (the Space key).
//
// Reference: moveTowardFeasibility, onedpath_ip.cpp:648-721 (F4: onedpath2_ip.cpp:536-609):
// collect the violated constraints (error > 0), form the Gram matrix a = G G^T of their
// gradients, m = a.colPivHouseholderQr().solve(err), dX = G^T (-m), var[0..2] += dX.
//
// At most 4 constraints can be violated at once in either variant (F3's come in -a-L / a-L
// pairs that exclude each other; F4 has 4), so the Gram matrix is at most 4x4 and the whole
// rank-revealing QR (ColPivHouseholderQR.h:480-611 restated, as in oracle/ip_oracle.c) runs
// in registers: every loop below is fully unrolled over NMAX = 4 with `< n` guards, and the
// runtime pivot column is applied through compare-and-select swaps, never through dynamic
// register indexing.
#pragma once

#include "ip_core.h"

namespace rp {

template <typename T> __device__ __forceinline__ T eps_();
template <> __device__ __forceinline__ double eps_<double>() { return 2.220446049250313e-16; }
template <> __device__ __forceinline__ float eps_<float>() { return 1.1920929e-07f; }
template <typename T> __device__ __forceinline__ T tiny_();
template <> __device__ __forceinline__ double tiny_<double>() { return 2.2250738585072014e-308; }
template <> __device__ __forceinline__ float tiny_<float>() { return 1.17549435e-38f; }

// x = colPivHouseholderQr(A).solve(b) for the leading n x n block, n <= 4.  A[r][c].
template <typename T>
__device__ __forceinline__ void colpiv_qr_solve4(int n, T (&A)[4][4], T (&b)[4], T (&x)[4])
{
    constexpr int N = 4;
    T nu[N], nd[N], tau[N];
    int perm[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        T s = T(0);
#pragma unroll
        for (int r = 0; r < N; ++r) if (r < n) s = fma_(A[r][c], A[r][c], s);
        nd[c] = nu[c] = sqrt_(s);
        perm[c] = c;
        tau[c] = T(0);
    }
    T maxn = T(0);
#pragma unroll
    for (int c = 0; c < N; ++c) if (c < n && nu[c] > maxn) maxn = nu[c];
    const T thr_helper = (maxn * eps_<T>()) * (maxn * eps_<T>()) / (T)n;
    const T down_thr = sqrt_(eps_<T>());
    int nz = n;

#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < n) {
            // pivot column: largest updated norm among k..n-1, first wins
            int big = k;
            T bigv = nu[k];
#pragma unroll
            for (int j = k + 1; j < N; ++j) if (j < n && nu[j] > bigv) { bigv = nu[j]; big = j; }
            if (nz == n && bigv * bigv < thr_helper * (T)(n - k)) nz = k;
#pragma unroll
            for (int j = k + 1; j < N; ++j) {
                const bool sw = (big == j);
#pragma unroll
                for (int r = 0; r < N; ++r) swap_if(sw, A[r][k], A[r][j]);
                swap_if(sw, nu[k], nu[j]);
                swap_if(sw, nd[k], nd[j]);
                const int pk = sw ? perm[j] : perm[k], pj = sw ? perm[k] : perm[j];
                perm[k] = pk;
                perm[j] = pj;
            }
            // Householder vector of rows k..n-1 of column k (Householder.h:65-94)
            T tail = T(0);
#pragma unroll
            for (int r = k + 1; r < N; ++r) if (r < n) tail = fma_(A[r][k], A[r][k], tail);
            const T c0 = A[k][k];
            T beta, tk;
            if (tail <= tiny_<T>()) {
                tk = T(0);
                beta = c0;
#pragma unroll
                for (int r = k + 1; r < N; ++r) if (r < n) A[r][k] = T(0);
            } else {
                beta = sqrt_(fma_(c0, c0, tail));
                if (c0 >= T(0)) beta = -beta;
                const T den = c0 - beta;
#pragma unroll
                for (int r = k + 1; r < N; ++r) if (r < n) A[r][k] = A[r][k] / den;
                tk = (beta - c0) / beta;
            }
            tau[k] = tk;
            A[k][k] = beta;
            // apply to the trailing columns (Householder.h:113-131) and to b (Q^T b on the fly:
            // _solve_impl applies H_0..H_{nz-1}; reflectors past nz are skipped below)
            if (tk != T(0)) {
#pragma unroll
                for (int j = k + 1; j < N; ++j) {
                    if (j < n) {
                        T t = A[k][j];
#pragma unroll
                        for (int r = k + 1; r < N; ++r) if (r < n) t = fma_(A[r][k], A[r][j], t);
                        A[k][j] = fma_(-tk, t, A[k][j]);
#pragma unroll
                        for (int r = k + 1; r < N; ++r) if (r < n) A[r][j] = fma_(-tk * A[r][k], t, A[r][j]);
                    }
                }
                if (k < nz) {
                    T t = b[k];
#pragma unroll
                    for (int r = k + 1; r < N; ++r) if (r < n) t = fma_(A[r][k], b[r], t);
                    b[k] = fma_(-tk, t, b[k]);
#pragma unroll
                    for (int r = k + 1; r < N; ++r) if (r < n) b[r] = fma_(-tk * A[r][k], t, b[r]);
                }
            } else if (n - k == 1 && k < nz) {
                b[k] *= T(1) - tk;
            }
            // column-norm downdate (ColPivHouseholderQR.h:551-571)
#pragma unroll
            for (int j = k + 1; j < N; ++j) {
                if (j < n && nu[j] != T(0)) {
                    T temp = abs_(A[k][j]) / nu[j];
                    temp = (T(1) + temp) * (T(1) - temp);
                    temp = temp < T(0) ? T(0) : temp;
                    const T ratio = nu[j] / nd[j];
                    const T temp2 = temp * (ratio * ratio);
                    if (temp2 <= down_thr) {
                        T s = T(0);
#pragma unroll
                        for (int r = k + 1; r < N; ++r) if (r < n) s = fma_(A[r][j], A[r][j], s);
                        nd[j] = nu[j] = sqrt_(s);
                    } else {
                        nu[j] *= sqrt_(temp);
                    }
                }
            }
        }
    }
    // back substitution on the leading nz x nz block, zeros for the rest, then un-permute
    T c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = (i < nz) ? b[i] : T(0);
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        if (i < nz) {
            c[i] = c[i] / A[i][i];
#pragma unroll
            for (int j = 0; j < i; ++j) c[j] = fma_(-c[i], A[j][i], c[j]);
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int d = 0; d < N; ++d) if (i < n && perm[i] == d) x[d] = c[i];
    }
}

// Returns false when nothing is violated (the reference then adds a zero dX).
template <typename T, int VARIANT>
__device__ __forceinline__ bool feasibility_move(const Acc<T> &e, T L, T &dxv, T &dx0, T &dx1)
{
    constexpr int NC = CMap<VARIANT>::NC;
    // compact the violated rows into at most 4 slots, in constraint order
    T g[4][3], er[4];
    int n = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) { g[s][0] = g[s][1] = g[s][2] = T(0); er[s] = T(0); }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const T ci = c_value<T, VARIANT>(i, e, L);
        if (ci > T(0)) {
            T gv, gt;
            c_grad<T, VARIANT>(i, e, gv, gt);
            const T g0 = gv, g1 = (c_segment<VARIANT>(i) == 0) ? gt : T(0), g2 = (c_segment<VARIANT>(i) == 0) ? T(0) : gt;
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
            A[r][c] = fma_(g[r][2], g[c][2], fma_(g[r][1], g[c][1], g[r][0] * g[c][0]));
    colpiv_qr_solve4<T>(n, A, er, m);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s < n) {
            dxv = fma_(g[s][0], -m[s], dxv);
            dx0 = fma_(g[s][1], -m[s], dx0);
            dx1 = fma_(g[s][2], -m[s], dx1);
        }
    }
    return true;
}

}  // namespace rp
