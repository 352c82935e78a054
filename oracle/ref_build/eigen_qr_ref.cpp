// eigen_qr_ref.cpp -- harness around the reference's OWN vendored Eigen 3.3.0.
//
// TEST INFRASTRUCTURE ONLY.  This TU is ours; the library it instantiates is the
// reference's, compiled from where it lies (/root/reference/libs/eigen, header-only,
// no stand-ins needed).  It exposes the exact call the hot path makes,
//     d = m.colPivHouseholderQr().solve(-r)          (onedpath_ip.cpp:886-887, 11x11)
//     d = m.colPivHouseholderQr().solve(-r)          (onedpath2_ip.cpp:774-775, 7x7)
//     m = a.colPivHouseholderQr().solve(err)         (onedpath_ip.cpp:693, dynamic n<=8)
// so that oracle/ip_oracle.c's restatement of ColPivHouseholderQR can be checked against
// the real thing on the KKT systems the step produces.  Output goes to oracle/_ref/ only.
#include <Eigen/Dense>

using namespace Eigen;

template <int N>
static int solve_fixed(const double *A, const double *b, double *x)
{
    Matrix<double, N, N> m = Map<const Matrix<double, N, N>>(A);   // column-major, as Eigen's default
    Matrix<double, N, 1> r = Map<const Matrix<double, N, 1>>(b);
    ColPivHouseholderQR<Matrix<double, N, N>> qr = m.colPivHouseholderQr();
    Matrix<double, N, 1> d = qr.solve(r);
    Map<Matrix<double, N, 1>> out(x);
    out = d;
    return (int)qr.nonzeroPivots();
}

extern "C" int ref_qr_solve(int n, const double *A, const double *b, double *x, int force_dynamic)
{
    if (!force_dynamic) {
        if (n == 11) return solve_fixed<11>(A, b, x);
        if (n == 7) return solve_fixed<7>(A, b, x);
    }
    MatrixXd m = Map<const MatrixXd>(A, n, n);
    VectorXd r = Map<const VectorXd>(b, n);
    ColPivHouseholderQR<MatrixXd> qr = m.colPivHouseholderQr();
    VectorXd d = qr.solve(r);
    Map<VectorXd> out(x, n);
    out = d;
    return (int)qr.nonzeroPivots();
}

// Matrix<double,N,1>::squaredNorm() as residualNorm calls it (onedpath_ip.cpp:791).
extern "C" double ref_squared_norm(int n, const double *v)
{
    if (n == 11) return Map<const Matrix<double, 11, 1>>(v).eval().squaredNorm();
    if (n == 7) return Map<const Matrix<double, 7, 1>>(v).eval().squaredNorm();
    return Map<const VectorXd>(v, n).squaredNorm();
}

extern "C" const char *ref_eigen_version()
{
#define RP_STR2(x) #x
#define RP_STR(x) RP_STR2(x)
    return RP_STR(EIGEN_WORLD_VERSION) "." RP_STR(EIGEN_MAJOR_VERSION) "." RP_STR(EIGEN_MINOR_VERSION);
}
