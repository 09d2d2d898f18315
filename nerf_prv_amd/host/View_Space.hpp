// View_Space.hpp -- candidate cameras of the planner.  Same class and member names as the
// reference's PRV_simulation/View_Space.hpp (View :40-199, View_Space :492-728); own
// implementation on a 60-line matrix helper instead of Eigen, no PCL/OctoMap members.
#pragma once
#include <memory>
#include <vector>

#include "Share_Data.hpp"
#include "mat4.hpp"

namespace prvhost {

class View {
public:
  Vec3 init_pos; // camera position in the world
  Mat4 pose;     // world -> camera of this view

  explicit View(Vec3 _init_pos) : init_pos(_init_pos), pose(Mat4::Identity()) {}

  // 0: roll closest to the previous camera's axes.  (Type 1, "y-top", has no live caller in
  // the reference -- View_Space.hpp:594 is commented out -- and is not provided.)
  // Restates View_Space.hpp:67-140.
  void get_next_camera_pos(const Mat4& now_camera_pose_world, const Vec3& object_center_world, int type_of_pose = 0) {
    (void)type_of_pose;
    const Mat4 cam_inv = now_camera_pose_world.inverse();
    const auto oc = cam_inv.mul(object_center_world.x, object_center_world.y, object_center_world.z, 1);
    const auto vc = cam_inv.mul(init_pos.x, init_pos.y, init_pos.z, 1);
    const Vec3 object(oc[0], oc[1], oc[2]), view(vc[0], vc[1], vc[2]);
    const Vec3 Z = (object - view).normalized();  // :79  camera looks at the object
    const Vec3 X = Z.cross(view).normalized();    // :81  degenerate when object == origin
    const Vec3 Y = Z.cross(X).normalized();       // :82
    Mat4 T = Mat4::Identity();
    T(0, 3) = -view.x;
    T(1, 3) = -view.y;
    T(2, 3) = -view.z;
    Mat4 R = Mat4::Identity();
    R(0, 0) = X.x; R(0, 1) = Y.x; R(0, 2) = Z.x;
    R(1, 0) = X.y; R(1, 1) = Y.y; R(1, 2) = Z.y;
    R(2, 0) = X.z; R(2, 1) = Y.z; R(2, 2) = Z.z;
    // roll search: 5 degree steps, keep the roll whose camera y (then x) axis image is
    // closest to the previous camera's; acos of an out-of-range value is NaN and loses
    // every comparison, exactly as in the reference (:101-128)
    Mat4 Rz_min = Mat4::Identity();
    Mat4 M = R.inverse() * T;
    double min_y = std::acos(M.mul(0, 1, 0, 1)[1]);
    double min_x = std::acos(M.mul(1, 0, 0, 1)[0]);
    for (double i = 5; i < 360; i += 5) {
      const double a = i * std::acos(-1.0) / 180.0;
      // Eigen builds this rotation through a quaternion (AngleAxis products), then a matrix
      const double qz = std::sin(a / 2), qw = std::cos(a / 2);
      const double tz = 2 * qz, twz = tz * qw, tzz = tz * qz;
      Mat4 Rz = Mat4::Identity();
      Rz(0, 0) = 1 - tzz; Rz(0, 1) = -twz;
      Rz(1, 0) = twz;     Rz(1, 1) = 1 - tzz;
      M = (R * Rz).inverse() * T;
      const double cos_y = std::acos(M.mul(0, 1, 0, 1)[1]);
      const double cos_x = std::acos(M.mul(1, 0, 0, 1)[0]);
      if (cos_y < min_y) {
        Rz_min = Rz; min_y = cos_y; min_x = cos_x;
      } else if (std::fabs(cos_y - min_y) < 1e-6 && cos_x < min_x) {
        Rz_min = Rz; min_y = cos_y; min_x = cos_x;
      }
    }
    pose = (R * Rz_min).inverse() * T; // :137
  }
};

class View_Space {
public:
  int num_of_views = 0;
  std::vector<View> views;
  Vec3 object_center_world;
  double predicted_size = 0.0;
  Mat4 now_camera_pose_world = Mat4::Identity();
  std::shared_ptr<Share_Data> share_data;

  // View_Space.hpp:517-558: centroid, 17/16 x bounding radius, hemisphere points scaled to
  // view_space_radius around the centroid, lower half dropped
  void get_view_space(const std::vector<Vec3>& points) {
    object_center_world = Vec3(0, 0, 0);
    for (const auto& p : points) object_center_world = object_center_world + p;
    const double n = (double)points.size();
    object_center_world = Vec3(object_center_world.x / n, object_center_world.y / n, object_center_world.z / n);
    predicted_size = 0.0;
    for (const auto& p : points) predicted_size = std::max(predicted_size, (object_center_world - p).norm());
    predicted_size *= 17.0 / 16.0;
    place_views();
  }

  // the same placement when centre and size are already known (synthetic scenes: no cloud)
  void set_view_space(const Vec3& center, double size) {
    object_center_world = center;
    predicted_size = size;
    place_views();
  }

  explicit View_Space(const std::shared_ptr<Share_Data>& _share_data) : share_data(_share_data) {
    num_of_views = share_data->num_of_views;
    now_camera_pose_world = Mat4::Identity(); // Share_Data.hpp:475; never updated by live code
  }

private:
  void place_views() {
    views.clear();
    for (size_t i = 0; i < share_data->pt_sphere.size(); i++) {
      if (share_data->pt_sphere[i][2] < 0) continue;
      const double scale = 1.0 / share_data->pt_norm * share_data->view_space_radius;
      views.emplace_back(Vec3(share_data->pt_sphere[i][0] * scale + object_center_world.x,
                              share_data->pt_sphere[i][1] * scale + object_center_world.y,
                              share_data->pt_sphere[i][2] * scale + object_center_world.z));
    }
  }
};

} // namespace prvhost
