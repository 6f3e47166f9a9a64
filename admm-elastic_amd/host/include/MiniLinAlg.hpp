// MiniLinAlg.hpp -- the few dense/sparse value types the reference's public API exposes through Eigen
// (VectorXd with segment<3>(), Vector3d, Vector3i/4i, Triplet<double>, a row-major sparse matrix).
// Eigen is neither installed here nor copyable from the reference tree, so the MI355X build carries its
// own minimal look-alikes in namespace admm.  Only what Solver / EnergyTerm users touch is provided.
#ifndef ADMM_MINILINALG_HPP
#define ADMM_MINILINALG_HPP 1

#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <vector>

#if defined(ADMM_WITH_EIGEN)
// Built against a real Eigen found at build time (-DADMM_WITH_EIGEN -I<eigen>; never copied into this tree): the value types of the public
// API ARE the Eigen types the reference's headers show (src/Solver.hpp:66-68, src/EnergyTerm.hpp:68-107), so code written against the
// reference -- its own samples/tests/test_lineartet.cpp -- compiles unchanged against this mirror (tests/cpp/build_reference_test.sh).
#include <Eigen/Dense>
#include <Eigen/Sparse>
namespace admm {
typedef Eigen::Vector3d Vec3;
typedef Eigen::Vector3i Vec3i;
typedef Eigen::Vector4i Vec4i;
typedef Eigen::VectorXd VecX;
typedef Eigen::Triplet<double> Triplet;
typedef Eigen::SparseMatrix<double, Eigen::RowMajor> SparseMat;
namespace la {      // the three things the mirror does with a sparse matrix beyond Eigen's own interface
inline void get_csr(const SparseMat &M, std::vector<int> &rp, std::vector<int> &ci, std::vector<double> &va) {
    SparseMat C = M; C.makeCompressed();
    rp.assign(C.outerIndexPtr(), C.outerIndexPtr() + C.rows() + 1); ci.assign(C.innerIndexPtr(), C.innerIndexPtr() + C.nonZeros()); va.assign(C.valuePtr(), C.valuePtr() + C.nonZeros());
}
inline void set_csr(SparseMat &M, int n, const std::vector<int> &rp, const std::vector<int> &ci, const std::vector<double> &va) {
    std::vector<Triplet> t; t.reserve(ci.size());
    for (int i = 0; i < n; ++i) for (int k = rp[i]; k < rp[i + 1]; ++k) t.emplace_back(i, ci[k], va[k]);
    M.resize(n, n); M.setFromTriplets(t.begin(), t.end());
}
inline void scale_columns(SparseMat &M, const VecX &s, double f) { M = (M * s.asDiagonal()).eval() * f; }      // M <- f M diag(s)
inline VecX zeros(std::size_t n) { return VecX::Zero((Eigen::Index)n); }
}
}
#else

namespace admm {

struct Vec3 {
    double v[3];
    Vec3() : v{0, 0, 0} {}
    Vec3(double x, double y, double z) : v{x, y, z} {}
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    Vec3 operator+(const Vec3 &o) const { return Vec3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec3 operator-(const Vec3 &o) const { return Vec3(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3 operator*(double s) const { return Vec3(v[0] * s, v[1] * s, v[2] * s); }
    Vec3 &operator+=(const Vec3 &o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
    Vec3 &operator-=(const Vec3 &o) { v[0] -= o.v[0]; v[1] -= o.v[1]; v[2] -= o.v[2]; return *this; }
    double dot(const Vec3 &o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    Vec3 cross(const Vec3 &o) const { return Vec3(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]); }
    double norm() const { return std::sqrt(dot(*this)); }
    const double *data() const { return v; }
};
inline Vec3 operator*(double s, const Vec3 &a) { return a * s; }

template <int N>
struct VecNi {
    int v[N];
    VecNi() { for (int i = 0; i < N; ++i) v[i] = 0; }
    int &operator[](int i) { return v[i]; }
    int operator[](int i) const { return v[i]; }
};
struct Vec3i : VecNi<3> { Vec3i() {} Vec3i(int a, int b, int c) { v[0] = a; v[1] = b; v[2] = c; } };
struct Vec4i : VecNi<4> { Vec4i() {} Vec4i(int a, int b, int c, int d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; } };

// Dense dynamic vector with the handful of Eigen::VectorXd operations Solver users rely on.
class VecX {
public:
    // assignable view of three consecutive entries: x.segment<3>(i) = Vec3(..);  Vec3 p = x.segment<3>(i);
    struct Seg3 {
        double *p;
        Seg3 &operator=(const Vec3 &a) { p[0] = a[0]; p[1] = a[1]; p[2] = a[2]; return *this; }
        operator Vec3() const { return Vec3(p[0], p[1], p[2]); }
        double &operator[](int i) { return p[i]; }
    };
    VecX() {}
    explicit VecX(std::size_t n, double fill = 0.0) : d_(n, fill) {}
    static VecX Zero(std::size_t n) { return VecX(n, 0.0); }
    static VecX Ones(std::size_t n) { return VecX(n, 1.0); }
    int rows() const { return (int)d_.size(); }
    int size() const { return (int)d_.size(); }
    void resize(std::size_t n) { d_.assign(n, 0.0); }
    void conservativeResize(std::size_t n) { d_.resize(n, 0.0); }
    void setZero() { for (double &x : d_) x = 0.0; }
    double &operator[](std::size_t i) { return d_[i]; }
    double operator[](std::size_t i) const { return d_[i]; }
    double *data() { return d_.data(); }
    const double *data() const { return d_.data(); }
    template <int N> Seg3 segment(int i) { static_assert(N == 3, "only segment<3> is provided"); return Seg3{d_.data() + i}; }
    template <int N> Vec3 segment(int i) const { static_assert(N == 3, "only segment<3> is provided"); return Vec3(d_[i], d_[i + 1], d_[i + 2]); }
    double norm() const { double s = 0; for (double x : d_) s += x * x; return std::sqrt(s); }
    VecX operator-(const VecX &o) const { VecX r(d_.size()); for (std::size_t i = 0; i < d_.size(); ++i) r.d_[i] = d_[i] - o.d_[i]; return r; }
    VecX operator+(const VecX &o) const { VecX r(d_.size()); for (std::size_t i = 0; i < d_.size(); ++i) r.d_[i] = d_[i] + o.d_[i]; return r; }
    std::vector<double> &std() { return d_; }
    const std::vector<double> &std() const { return d_; }
private:
    std::vector<double> d_;
};

struct Triplet {
    int r, c; double val;
    Triplet(int r_, int c_, double v_) : r(r_), c(c_), val(v_) {}
    int row() const { return r; }
    int col() const { return c; }
    double value() const { return val; }
};

// Row-major sparse matrix (CSR, duplicates summed) -- stands where the reference API shows
// Eigen::SparseMatrix<double,RowMajor>.
class SparseMat {
public:
    SparseMat() : rows_(0), cols_(0), rowptr_(1, 0) {}
    void resize(int r, int c) { rows_ = r; cols_ = c; rowptr_.assign(r + 1, 0); col_.clear(); val_.clear(); }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    int nonZeros() const { return (int)col_.size(); }
    template <class It> void setFromTriplets(It b, It e) {
        std::vector<std::vector<std::pair<int, double> > > rw(rows_);
        for (It t = b; t != e; ++t) {
            if (t->row() < 0 || t->row() >= rows_ || t->col() < 0 || t->col() >= cols_) throw std::runtime_error("SparseMat: triplet out of range");
            std::vector<std::pair<int, double> > &r = rw[t->row()];
            bool found = false;
            for (auto &p : r) if (p.first == t->col()) { p.second += t->value(); found = true; break; }
            if (!found) r.push_back(std::make_pair(t->col(), t->value()));
        }
        rowptr_.assign(rows_ + 1, 0); col_.clear(); val_.clear();
        for (int i = 0; i < rows_; ++i) {
            for (auto &p : rw[i]) { col_.push_back(p.first); val_.push_back(p.second); }
            rowptr_[i + 1] = (int)col_.size();
        }
    }
    VecX operator*(const VecX &x) const {
        VecX y(rows_);
        for (int i = 0; i < rows_; ++i) { double a = 0; for (int k = rowptr_[i]; k < rowptr_[i + 1]; ++k) a += val_[k] * x[col_[k]]; y[i] = a; }
        return y;
    }
    const std::vector<int> &rowptr() const { return rowptr_; }
    const std::vector<int> &colind() const { return col_; }
    const std::vector<double> &values() const { return val_; }
    void setCsr(int n, const std::vector<int> &rp, const std::vector<int> &ci, const std::vector<double> &va) { rows_ = cols_ = n; rowptr_ = rp; col_ = ci; val_ = va; }
    SparseMat transpose() const {          // counting sort by column: rows of the result keep increasing column order
        SparseMat t; t.rows_ = cols_; t.cols_ = rows_; t.rowptr_.assign(cols_ + 1, 0);
        for (int c : col_) t.rowptr_[c + 1] += 1;
        for (int i = 0; i < cols_; ++i) t.rowptr_[i + 1] += t.rowptr_[i];
        t.col_.resize(col_.size()); t.val_.resize(val_.size());
        std::vector<int> pos(t.rowptr_.begin(), t.rowptr_.end() - 1);
        for (int i = 0; i < rows_; ++i)
            for (int k = rowptr_[i]; k < rowptr_[i + 1]; ++k) { const int o = pos[col_[k]]++; t.col_[o] = i; t.val_[o] = val_[k]; }
        return t;
    }
    void scaleColumns(const VecX &s, double f) { for (size_t k = 0; k < col_.size(); ++k) val_[k] *= f * s[col_[k]]; }   // this <- f * this * diag(s)
    double coeff(int r, int c) const { for (int k = rowptr_[r]; k < rowptr_[r + 1]; ++k) if (col_[k] == c) return val_[k]; return 0.0; }
private:
    int rows_, cols_;
    std::vector<int> rowptr_, col_;
    std::vector<double> val_;
};

namespace la {
inline void get_csr(const SparseMat &M, std::vector<int> &rp, std::vector<int> &ci, std::vector<double> &va) { rp = M.rowptr(); ci = M.colind(); va = M.values(); }
inline void set_csr(SparseMat &M, int n, const std::vector<int> &rp, const std::vector<int> &ci, const std::vector<double> &va) { M.setCsr(n, rp, ci, va); }
inline void scale_columns(SparseMat &M, const VecX &s, double f) { M.scaleColumns(s, f); }
inline VecX zeros(std::size_t n) { return VecX::Zero(n); }
}

} // namespace admm
#endif
#endif
