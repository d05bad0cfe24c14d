// Complex element type of the device tensors (TenElemT = QLTEN_Complex = std::complex<double> in the reference:
// every hot-path test of the reference is compiled for QLTEN_Double and QLTEN_Complex, tests/CMakeLists.txt:57-100).
// A plain struct with the arithmetic the kernels use, usable on host and device; layout = interleaved (re, im), the
// payload layout of a complex .qlten file and of std::complex.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace pepsgpu {

template <typename R>
struct cplx {
  R re, im;
  cplx() = default;                       // trivial: usable in __shared__ arrays (no initialisation)
  __host__ __device__ cplx(R r) : re(r), im(R(0)) {}
  __host__ __device__ cplx(R r, R i) : re(r), im(i) {}
  template <typename S>
  __host__ __device__ explicit cplx(const cplx<S> &o) : re(R(o.re)), im(R(o.im)) {}
  __host__ __device__ explicit cplx(int v) : re(R(v)), im(R(0)) {}
  __host__ __device__ cplx &operator+=(const cplx &o) { re += o.re; im += o.im; return *this; }
  __host__ __device__ cplx &operator-=(const cplx &o) { re -= o.re; im -= o.im; return *this; }
  __host__ __device__ cplx &operator*=(const cplx &o) { const R r = re * o.re - im * o.im; im = re * o.im + im * o.re; re = r; return *this; }
  __host__ __device__ cplx &operator*=(R s) { re *= s; im *= s; return *this; }
};
template <typename R> __host__ __device__ inline cplx<R> operator+(cplx<R> a, const cplx<R> &b) { return a += b; }
template <typename R> __host__ __device__ inline cplx<R> operator-(cplx<R> a, const cplx<R> &b) { return a -= b; }
template <typename R> __host__ __device__ inline cplx<R> operator*(const cplx<R> &a, const cplx<R> &b) {
  return cplx<R>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <typename R> __host__ __device__ inline cplx<R> operator*(const cplx<R> &a, R s) { return cplx<R>(a.re * s, a.im * s); }
template <typename R> __host__ __device__ inline cplx<R> operator*(R s, const cplx<R> &a) { return cplx<R>(a.re * s, a.im * s); }
template <typename R> __host__ __device__ inline cplx<R> operator-(const cplx<R> &a) { return cplx<R>(-a.re, -a.im); }

typedef cplx<double> c128;

template <typename T> struct is_cplx : std::false_type {};
template <typename R> struct is_cplx<cplx<R>> : std::true_type {};
template <typename T> struct real_of { typedef T type; };
template <typename R> struct real_of<cplx<R>> { typedef R type; };
// accumulation type of the Gram matrices and of the final inner products: float64, real or complex
template <typename T> struct acc64_of { typedef double type; };
template <typename R> struct acc64_of<cplx<R>> { typedef cplx<double> type; };

// conj / |x|^2 / real part, uniform over real and complex element types
__host__ __device__ inline float conj_of(float x) { return x; }
__host__ __device__ inline double conj_of(double x) { return x; }
template <typename R> __host__ __device__ inline cplx<R> conj_of(const cplx<R> &x) { return cplx<R>(x.re, -x.im); }
__host__ __device__ inline double abs2_of(float x) { return (double)x * (double)x; }
__host__ __device__ inline double abs2_of(double x) { return x * x; }
template <typename R> __host__ __device__ inline double abs2_of(const cplx<R> &x) { return (double)x.re * (double)x.re + (double)x.im * (double)x.im; }
__host__ __device__ inline double real_part(float x) { return x; }
__host__ __device__ inline double real_part(double x) { return x; }
template <typename R> __host__ __device__ inline double real_part(const cplx<R> &x) { return (double)x.re; }
// x * s with a real scale factor given as double
__host__ __device__ inline float scaled(float x, double s) { return (float)((double)x * s); }
__host__ __device__ inline double scaled(double x, double s) { return x * s; }
template <typename R> __host__ __device__ inline cplx<R> scaled(const cplx<R> &x, double s) { return cplx<R>(R(x.re * s), R(x.im * s)); }

}  // namespace pepsgpu
