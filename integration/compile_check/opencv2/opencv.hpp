// Syntax-check stand-in (integration/compile_check only): common/pinhole_camera_impl.h includes <opencv2/opencv.hpp>
// for PinholeCamera::FromFile (cv::FileStorage), which the adapter never calls.  OpenCV is not in the build image; this
// header lets `hipcc -fsyntax-only integration/sage_adapter.cpp` see the reference's REAL declarations of
// df::PinholeCamera / df::CameraPyramid and of the seven kernel entry points.  It is not part of the product, of the
// oracle, or of any parity claim.
#pragma once
#include <string>
namespace cv
{
struct FileNode
{
  template <class T>
  void operator>>(T &) const {}
  bool empty() const { return true; }
};
struct Mat
{
  template <class T>
  T at(int, int) const { return T(); }
};
struct FileStorage
{
  enum { READ = 0 };
  FileStorage(const std::string &, int) {}
  bool isOpened() const { return false; }
  FileNode operator[](const char *) const { return FileNode(); }
};
} // namespace cv
