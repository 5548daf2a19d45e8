/*
 * mpc_planner_types/data_types.h -- the data types that cross the accelerated path, restated after the reference's
 * mpc_planner_types/include/mpc_planner_types/data_types.h:13-134 (same names, members and constructors) on the 2-double
 * Vector2d of solver_interface.h (Eigen is not in the build image).  Types the path never touches (ReferencePath bounds,
 * FixedSizeTrajectory, costmap) are left to the reference.
 */
#ifndef MPC_DATA_TYPES_HIP_H
#define MPC_DATA_TYPES_HIP_H

#include <cmath>
#include <vector>

#include <mpc_planner_solver/solver_interface.h>

namespace MPCPlanner
{
    struct Disc                                                     /* data_types.h:13-22, data_types.cpp */
    {
        double offset, radius;
        Disc(const double offset_, const double radius_) : offset(offset_), radius(radius_) {}
        Vector2d getPosition(const Vector2d &robot_position, const double angle) const
        {
            return Vector2d(robot_position(0) + offset * std::cos(angle), robot_position(1) + offset * std::sin(angle));
        }
        Vector2d toRobotCenter(const Vector2d &disc_position, const double angle) const
        {
            return Vector2d(disc_position(0) - offset * std::cos(angle), disc_position(1) - offset * std::sin(angle));
        }
    };

    struct Halfspace                                                /* :24-31  A x <= b */
    {
        Vector2d A;
        double b;
        Halfspace(const Vector2d &A_, const double b_) : A(A_), b(b_) {}
    };
    typedef std::vector<Halfspace> StaticObstacle;                  /* for one k: the halfspaces of the free-space polytope */

    enum class PredictionType { DETERMINISTIC = 0, GAUSSIAN, NONGAUSSIAN, NONE };     /* :34-40 */

    struct PredictionStep                                           /* :42-54 */
    {
        Vector2d position;
        double angle;
        double major_radius, minor_radius;
        PredictionStep(const Vector2d &position_, double angle_, double major_radius_, double minor_radius_)
            : position(position_), angle(angle_), major_radius(major_radius_), minor_radius(minor_radius_) {}
    };
    typedef std::vector<PredictionStep> Mode;

    struct Prediction                                               /* :58-70 */
    {
        PredictionType type;
        std::vector<Mode> modes;
        std::vector<double> probabilities;
        Prediction() : type(PredictionType::NONE) {}
        Prediction(PredictionType type_) : type(type_)
        {
            if (type == PredictionType::DETERMINISTIC || type == PredictionType::GAUSSIAN) { modes.emplace_back(); probabilities.emplace_back(1.); }
        }
        bool empty() const { return modes.empty() || (modes.size() > 0 && modes[0].empty()); }
    };

    enum class ObstacleType { STATIC = 0, DYNAMIC };                /* :72-76 */

    struct DynamicObstacle                                          /* :78-91 */
    {
        int index;
        Vector2d position;
        double angle;
        double radius;
        ObstacleType type{ObstacleType::DYNAMIC};
        Prediction prediction;
        DynamicObstacle(int _index, const Vector2d &_position, double _angle, double _radius, ObstacleType _type = ObstacleType::DYNAMIC)
            : index(_index), position(_position), angle(_angle), radius(_radius), type(_type) {}
    };

    /* One cubic segment of the contouring reference (contouring.cpp:94-124 reads these numbers out of RosTools::Spline2D, which
     * is not in the reference tree): x(t) = ax t^3 + bx t^2 + cx t + dx, same for y, t = s - start. */
    struct PathSegment { double ax, bx, cx, dx, ay, by, cy, dy, start; };

    /* What GuidanceConstraints reads from the (external) guidance_planner per trajectory (guidance_constraints.cpp:340-360,
     * 390-414): position / velocity of its spline at t = k dt, k = 0..N, the topology class and the selection flag. */
    struct GuidanceTrajectory
    {
        std::vector<Vector2d> positions, velocities;
        int topology_class{0};
        bool previously_selected{false};
        int color{0};
    };
}
#endif
