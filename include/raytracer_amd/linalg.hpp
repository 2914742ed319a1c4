// Host-side vector / matrix types of the kept API (reference: include/raytracer/linalg/{Vector3f,Vector2f,Matrix3f}.cuh).
// A caller of the reference builds scenes and cameras with these, so the mirror offers the same members with the same
// arithmetic: scalar division multiplies by a reciprocal formed in DOUBLE and rounded to float once (Vector3f.cuh:106-109,
// :133-135, :161-164), Matrix3f stores columns and indexes (row, column) (Matrix3f.cuh:165-171), M * v sums x*col0 + y*col1 +
// z*col2 in that order (:243-247).  Host only: the device side of the render path has its own math (csrc/device_math.hpp).
// Not reproduced: `outer()` of the reference writes nine doubles into a matrix of nine floats (Matrix3f.cuh:261-276); here it
// writes floats.
#pragma once
#include <cmath>

struct Vector3f {
  union { struct { float x, y, z; }; struct { float r, g, b; }; };
  Vector3f() : x(0), y(0), z(0) {}
  Vector3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  Vector3f(float c) : x(c), y(c), z(c) {}                      // NOLINT: implicit, as in the reference (Vector3f.cuh:41)
  float dot(const Vector3f &v) const { return v.x * x + v.y * y + v.z * z; }
  Vector3f cross(const Vector3f &v) const { return {y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x}; }
  float &operator[](int i) { return (&x)[i]; }
  const float &operator[](int i) const { return (&x)[i]; }
  bool operator==(const Vector3f &v) const { return v.x == x && v.y == y && v.z == z; }
  Vector3f operator-() const { return {-x, -y, -z}; }
  Vector3f operator+(const Vector3f &v) const { return {x + v.x, y + v.y, z + v.z}; }
  Vector3f operator-(const Vector3f &v) const { return {x - v.x, y - v.y, z - v.z}; }
  Vector3f operator*(const Vector3f &v) const { return {x * v.x, y * v.y, z * v.z}; }
  Vector3f operator/(const Vector3f &v) const { return {x / v.x, y / v.y, z / v.z}; }
  Vector3f operator*(const float &c) const { return {x * c, y * c, z * c}; }
  Vector3f operator/(const float &c) const {
    const float rc = (float) (1.0 / (double) c);
    return {rc * x, rc * y, rc * z};
  }
  void operator+=(const Vector3f &v) { x += v.x; y += v.y; z += v.z; }
  void operator-=(const Vector3f &v) { x -= v.x; y -= v.y; z -= v.z; }
  void operator*=(const float &c) { x *= c; y *= c; z *= c; }
  void operator/=(const float &c) { (*this) *= (float) (1. / (double) c); }      // the double reciprocal binds to `const float &`
  Vector3f rcp() const { return {(float) (1.0 / (double) x), (float) (1.0 / (double) y), (float) (1.0 / (double) z)}; }
  float norm() const { return (float) std::sqrt((double) (x * x + y * y + z * z)); }
  float norm2() const { return x * x + y * y + z * z; }
  Vector3f unit() const {
    const float rn = (float) (1. / (double) norm());
    return (*this) * rn;
  }
  void normalize() { (*this) /= norm(); }
  float illum() const { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
};
inline Vector3f operator*(const float &c, const Vector3f &v) { return {c * v.x, c * v.y, c * v.z}; }
inline Vector3f operator/(const float &c, const Vector3f &v) { return {c / v.x, c / v.y, c / v.z}; }

struct Vector2f {
  float x, y;
  Vector2f() : x(0), y(0) {}
  Vector2f(float x_, float y_) : x(x_), y(y_) {}
  float dot(const Vector2f &v) const { return x * v.x + y * v.y; }
  float cross(const Vector2f &v) const { return x * v.y - y * v.x; }
  float &operator[](int i) { return (&x)[i]; }
  const float &operator[](int i) const { return (&x)[i]; }
  Vector2f operator-() const { return {-x, -y}; }
  Vector2f operator+(const Vector2f &v) const { return {x + v.x, y + v.y}; }
  Vector2f operator-(const Vector2f &v) const { return {x - v.x, y - v.y}; }
  Vector2f operator*(float c) const { return {x * c, y * c}; }
  Vector2f operator/(float c) const { return {x / c, y / c}; }   // true division here (Vector2f.cuh:105-108), unlike Vector3f
  void operator+=(const Vector2f &v) { x += v.x; y += v.y; }
  void operator-=(const Vector2f &v) { x -= v.x; y -= v.y; }
  void operator*=(float c) { x *= c; y *= c; }
  void operator/=(float c) { x /= c; y /= c; }
  float norm() const { return (float) std::sqrt((double) (x * x + y * y)); }
  float norm2() const { return x * x + y * y; }
  Vector2f unit() const { return *this / norm(); }
};
inline Vector2f operator*(float c, const Vector2f &v) { return {c * v.x, c * v.y}; }

struct Vector3i {
  int x = 0, y = 0, z = 0;
};

class Matrix3f {
protected:
  Vector3f entries[3];                                         // columns

public:
  Matrix3f() {                                                 // identity (Matrix3f.cuh:16-22)
    for (int j = 0; j < 3; j++) entries[j] = Vector3f(j == 0, j == 1, j == 2);
  }
  explicit Matrix3f(const float *row_major) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) (*this)(i, j) = row_major[i * 3 + j];
  }
  Matrix3f(float m00, float m01, float m02, float m10, float m11, float m12, float m20, float m21, float m22) {
    const float rows[9] = {m00, m01, m02, m10, m11, m12, m20, m21, m22};
    *this = Matrix3f(rows);
  }
  float &operator()(int i, int j) { return entries[j][i]; }
  const float &operator()(int i, int j) const { return entries[j][i]; }
  Vector3f &operator[](int j) { return entries[j]; }
  const Vector3f &operator[](int j) const { return entries[j]; }
  Vector3f &column(int j) { return entries[j]; }
  const Vector3f &column(int j) const { return entries[j]; }
  void zero(float val = 0.0f) { entries[0] = entries[1] = entries[2] = Vector3f(val, val, val); }
  static Matrix3f identity() { return Matrix3f(); }
  static Matrix3f crossProduct(const Vector3f &u) { return {0, -u.z, u.y, u.z, 0, -u.x, -u.y, u.x, 0}; }
  float det() const {
    const Matrix3f &A = *this;
    return -A(0, 2) * A(1, 1) * A(2, 0) + A(0, 1) * A(1, 2) * A(2, 0) + A(0, 2) * A(1, 0) * A(2, 1) - A(0, 0) * A(1, 2) * A(2, 1) -
           A(0, 1) * A(1, 0) * A(2, 2) + A(0, 0) * A(1, 1) * A(2, 2);
  }
  float norm() const { return (float) std::sqrt((double) (entries[0].norm2() + entries[1].norm2() + entries[2].norm2())); }
  Matrix3f T() const {
    Matrix3f B;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) B(i, j) = (*this)(j, i);
    return B;
  }
  Matrix3f inv() const {                                       // adjugate / determinant, cofactors as Matrix3f.cuh:149-157 orders them
    const Matrix3f &A = *this;
    Matrix3f B(-A(1, 2) * A(2, 1) + A(1, 1) * A(2, 2), A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2), -A(0, 2) * A(1, 1) + A(0, 1) * A(1, 2),
               A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2), -A(0, 2) * A(2, 0) + A(0, 0) * A(2, 2), A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2),
               -A(1, 1) * A(2, 0) + A(1, 0) * A(2, 1), A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1), -A(0, 1) * A(1, 0) + A(0, 0) * A(1, 1));
    B /= det();
    return B;
  }
  void operator+=(const Matrix3f &B) { for (int j = 0; j < 3; j++) entries[j] += B[j]; }
  void operator/=(float c) {
    const float rc = (float) (1. / (double) c);                // (:252-256: the double reciprocal is narrowed by Vector3f::operator*=)
    for (int j = 0; j < 3; j++) entries[j] *= rc;
  }
  Matrix3f operator-() const { Matrix3f B; for (int j = 0; j < 3; j++) B[j] = -entries[j]; return B; }
  Matrix3f operator-(const Matrix3f &B) const { Matrix3f C; for (int j = 0; j < 3; j++) C[j] = entries[j] - B[j]; return C; }
  Matrix3f operator*(float c) const { Matrix3f B; for (int j = 0; j < 3; j++) B[j] = entries[j] * c; return B; }
  Vector3f operator*(const Vector3f &v) const { return v.x * entries[0] + v.y * entries[1] + v.z * entries[2]; }
  Matrix3f operator*(const Matrix3f &B) const { Matrix3f C; for (int j = 0; j < 3; j++) C[j] = (*this) * B[j]; return C; }
};
inline Matrix3f operator*(float c, const Matrix3f &A) { return A * c; }
inline Matrix3f outer(const Vector3f &u, const Vector3f &v) {
  Matrix3f B;
  for (int j = 0; j < 3; j++) B[j] = u * v[j];
  return B;
}
