// TEST STAND-IN (never shipped): the member NAMES and types of ModelParameters / RawResults of the reference's
// src/miqp_planner_data.hpp:46-185 (and of the parameter structs it embeds), which is what include/cplex_wrapper.hpp is
// written against.  Inside a reference checkout the real header is used instead.
#pragma once
#include <vector>
#include <Eigen/Dense>
#include <unsupported/Eigen/CXX11/Tensor>
namespace miqp { namespace planner {
struct LimitPerRegionParameters { Eigen::MatrixXd min_x, max_x, min_y, max_y; };
struct PolynomialCurvatureParameters { Eigen::MatrixXd POLY_KAPPA_AX_MAX, POLY_KAPPA_AX_MIN; };
struct PolynomialOrientationParameters { Eigen::MatrixXd POLY_SINT_UB, POLY_SINT_LB, POLY_COSS_UB, POLY_COSS_LB; };
typedef Eigen::MatrixXd FractionParameters;
struct RawResults {
  Eigen::Tensor<double, 2> u_x, u_y, pos_x, vel_x, acc_x, pos_y, vel_y, acc_y, pos_x_front_UB, pos_x_front_LB, pos_y_front_UB, pos_y_front_LB;
  Eigen::Tensor<int, 3> notWithinEnvironmentRear, notWithinEnvironmentFrontUbUb, notWithinEnvironmentFrontLbUb, notWithinEnvironmentFrontUbLb,
      notWithinEnvironmentFrontLbLb, active_region;
  Eigen::Tensor<int, 2> region_change_not_allowed_x_positive, region_change_not_allowed_y_positive, region_change_not_allowed_x_negative,
      region_change_not_allowed_y_negative, region_change_not_allowed_combined;
  Eigen::Tensor<int, 4> deltacc; Eigen::Tensor<int, 5> deltacc_front; Eigen::Tensor<int, 4> car2car_collision, slackvars;
  Eigen::Tensor<int, 3> slackvarsObstacle; Eigen::Tensor<int, 4> slackvarsObstacle_front;
  int N = 0, NrEnvironments = 0, NrRegions = 0, NrObstacles = 0, MaxLinesObstacles = 0, NrCarToCarCollisions = 0, NrCars = 0;
};
struct ModelParameters {
  float max_solution_time = 10, relative_mip_gap_tolerance = 0.1f; int mipdisplay = 0, mipemphasis = 0; float relobjdif = 0;
  int cutpass = 0, probe = 0, repairtries = 0, rinsheur = 0, varsel = 0, mircuts = 0, parallelmode = 0;
  int NumSteps = 0; float ts = 0; int nr_regions = 0, NumCars = 0;
  float min_vel_x_y = 0, max_vel_x_y = 0, total_min_acc = 0, total_max_acc = 0, total_min_jerk = 0, total_max_jerk = 0;
  Eigen::VectorXd agent_safety_distance, agent_safety_distance_slack; float maximum_slack = 0;
  Eigen::VectorXd WEIGHTS_POS_X, WEIGHTS_VEL_X, WEIGHTS_ACC_X, WEIGHTS_POS_Y, WEIGHTS_VEL_Y, WEIGHTS_ACC_Y, WEIGHTS_JERK_X, WEIGHTS_JERK_Y;
  float WEIGHTS_SLACK = 0, WEIGHTS_SLACK_OBSTACLE = 0;
  Eigen::VectorXd WheelBase, CollisionRadius, BufferReference;
  Eigen::MatrixXd IntitialState, x_ref, vx_ref, y_ref, vy_ref;
  LimitPerRegionParameters acc_limit_params, jerk_limit_params;
  Eigen::VectorXi initial_region; Eigen::MatrixXi possible_region;
  int nr_obstacles = 0; std::vector<std::vector<Eigen::MatrixXd>> ObstacleConvexPolygon; int max_lines_obstacles = 0;
  std::vector<int> obstacle_is_soft; int nr_environments = 0; std::vector<Eigen::MatrixXd> MultiEnvironmentConvexPolygon;
  FractionParameters fraction_parameters; float minimum_region_change_speed = 0;
  PolynomialCurvatureParameters poly_curvature_params; PolynomialOrientationParameters poly_orientation_params;
};
} }  // namespace miqp::planner
