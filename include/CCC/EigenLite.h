/* EigenLite.h -- the handful of Eigen / ForceColl types the header shims of this repo need, for builds where Eigen3
 * and ForceControlCollection are not installed (they are not in the MI355X build image).  With Eigen available
 * (__has_include(<Eigen/Core>)) the real types are used and this file only provides the Contact stand-in.
 */
#pragma once

#include <array>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#if __has_include(<Eigen/Core>)
#  include <Eigen/Core>
namespace CCC
{
using Vector2d = Eigen::Vector2d;
using Vector3d = Eigen::Vector3d;
using Matrix3d = Eigen::Matrix3d;
using VectorXd = Eigen::VectorXd;
} // namespace CCC
#else
namespace CCC
{
struct Vector2d
{
  double v[2] = {0.0, 0.0};
  Vector2d() = default;
  Vector2d(double x, double y) : v{x, y} {}
  static Vector2d Zero()
  {
    return Vector2d();
  }
  double & x()
  {
    return v[0];
  }
  double & y()
  {
    return v[1];
  }
  double x() const
  {
    return v[0];
  }
  double y() const
  {
    return v[1];
  }
  double & operator[](int i)
  {
    return v[i];
  }
  double operator[](int i) const
  {
    return v[i];
  }
};

struct Vector3d
{
  double v[3] = {0.0, 0.0, 0.0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  static Vector3d Zero()
  {
    return Vector3d();
  }
  static Vector3d Constant(double c)
  {
    return Vector3d(c, c, c);
  }
  double & x()
  {
    return v[0];
  }
  double & y()
  {
    return v[1];
  }
  double & z()
  {
    return v[2];
  }
  double x() const
  {
    return v[0];
  }
  double y() const
  {
    return v[1];
  }
  double z() const
  {
    return v[2];
  }
  double & operator[](int i)
  {
    return v[i];
  }
  double operator[](int i) const
  {
    return v[i];
  }
};

/** Row-major 3x3 matrix, operator()(r, c) like Eigen. */
struct Matrix3d
{
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  static Matrix3d Identity()
  {
    return Matrix3d();
  }
  double & operator()(int r, int c)
  {
    return m[r * 3 + c];
  }
  double operator()(int r, int c) const
  {
    return m[r * 3 + c];
  }
};

/** Dynamic vector with the subset of Eigen::VectorXd the shims return / accept. */
struct VectorXd
{
  std::vector<double> v;
  VectorXd() = default;
  explicit VectorXd(int n) : v(static_cast<size_t>(n), 0.0) {}
  static VectorXd Zero(int n)
  {
    return VectorXd(n);
  }
  int size() const
  {
    return static_cast<int>(v.size());
  }
  void setZero(int n)
  {
    v.assign(static_cast<size_t>(n), 0.0);
  }
  double & operator[](int i)
  {
    return v[static_cast<size_t>(i)];
  }
  double operator[](int i) const
  {
    return v[static_cast<size_t>(i)];
  }
  double * data()
  {
    return v.data();
  }
  const double * data() const
  {
    return v.data();
  }
};
} // namespace CCC
#endif

#if __has_include(<ForceColl/Contact.h>)
#  include <ForceColl/Contact.h>
namespace CCC
{
using Contact = ForceColl::Contact;
}
#else
namespace CCC
{
/** Stand-in for ForceColl::Contact (external to the reference): the data the DDP path reads from it,
    /root/reference/src/DdpCentroidal.cpp:25-28,49-60 -- vertexWithRidgeList_ and ridgeNum(). */
struct Contact
{
  struct VertexWithRidge
  {
    Vector3d vertex;
    std::vector<Vector3d> ridgeList;
  };
  std::string name_;
  std::vector<VertexWithRidge> vertexWithRidgeList_;

  int ridgeNum() const
  {
    int n = 0;
    for(const auto & vr : vertexWithRidgeList_) n += static_cast<int>(vr.ridgeList.size());
    return n;
  }
};

/** makeContactFromRect of /root/reference/tests/src/ContactManager.h:10-21 with the friction-pyramid convention of
    this repo (4 ridges per vertex, normalize([mu cos th, mu sin th, 1]), identity pose). */
inline std::shared_ptr<Contact> makeContactFromRect(const std::array<Vector2d, 2> & rect_min_max, double mu = 0.5)
{
  auto c = std::make_shared<Contact>();
  c->name_ = "ContactFromRect";
  const double vx[4] = {rect_min_max[0].x(), rect_min_max[0].x(), rect_min_max[1].x(), rect_min_max[1].x()};
  const double vy[4] = {rect_min_max[0].y(), rect_min_max[1].y(), rect_min_max[1].y(), rect_min_max[0].y()};
  const double pi = 3.14159265358979323846;
  for(int k = 0; k < 4; k++)
  {
    Contact::VertexWithRidge vr;
    vr.vertex = Vector3d(vx[k], vy[k], 0.0);
    for(int i = 0; i < 4; i++)
    {
      const double th = 2.0 * pi * i / 4;
      const double r[3] = {mu * std::cos(th), mu * std::sin(th), 1.0};
      const double nrm = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      vr.ridgeList.push_back(Vector3d(r[0] / nrm, r[1] / nrm, r[2] / nrm));
    }
    c->vertexWithRidgeList_.push_back(vr);
  }
  return c;
}
} // namespace CCC
#endif
