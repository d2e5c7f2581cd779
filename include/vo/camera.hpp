// vo/camera.hpp -- the reference's Camera (camera.h:12-63, camera.cpp) over
// libvo_hip.so.  Same constructor defaults, same accessors, same semantics of
// projectPoint / projectPoints (int depth bounds inclusive, u against cols-1,
// v against rows-1, (-1,-1) for invalid points, stable compaction).
#pragma once

#include "context.hpp"
#include "types.hpp"

namespace vo {

class Camera {
 public:
  Camera(int rows = 100, int cols = 100, int z_near = 0, int z_far = 10,
         const Matrix3f& camera_matrix = Matrix3f::Identity(),
         const Isometry3f& world_in_camera_pose = Isometry3f::Identity())
      : _rows(rows), _cols(cols), _z_near(z_near), _z_far(z_far), _camera_matrix(camera_matrix),
        _world_in_camera_pose(world_in_camera_pose) {}

  //! projects a single point (camera.h:25-37); false if outside the depth range or the image
  bool projectPoint(Vector2f& image_point, const Vector3f& world_point) const {
    int n_out = 0, n_inside = 0;
    check(vo_project_points(default_context().handle(), _rows, _cols, _z_near, _z_far, _camera_matrix.data(),
                            _world_in_camera_pose.data(), world_point.data(), 1, 1, image_point.data(), &n_out,
                            &n_inside), "Camera::projectPoint");
    return n_inside == 1;
  }

  //! projects a bunch of world points (camera.cpp:16-37); returns the number inside
  int projectPoints(Vector2fVector& image_points, const Vector3fVector& world_points, bool keep_indices = false) const {
    const int n = static_cast<int>(world_points.size());
    image_points.resize(world_points.size());
    int n_out = 0, n_inside = 0;
    check(vo_project_points(default_context().handle(), _rows, _cols, _z_near, _z_far, _camera_matrix.data(),
                            _world_in_camera_pose.data(), n ? world_points[0].data() : nullptr, n,
                            keep_indices ? 1 : 0, n ? image_points[0].data() : nullptr, &n_out, &n_inside),
          "Camera::projectPoints");
    image_points.resize(static_cast<size_t>(n_out));
    return n_inside;
  }

  const Isometry3f& worldInCameraPose() const { return _world_in_camera_pose; }
  void setWorldInCameraPose(const Isometry3f& pose) { _world_in_camera_pose = pose; }
  const Matrix3f& cameraMatrix() const { return _camera_matrix; }
  int rows() const { return _rows; }
  int cols() const { return _cols; }
  int zNear() const { return _z_near; }
  int zFar() const { return _z_far; }

 protected:
  int _rows, _cols, _z_near, _z_far;   // ints, as in camera.h:56-59
  Matrix3f _camera_matrix;
  Isometry3f _world_in_camera_pose;
};

}  // namespace vo
