"""MI355X-native batched NMPC solver: drop-in for the reference's solver(x0,p,...) / shift() path."""
from .problem import (ProblemConfig, centralized_one_robot, centralized_two_robots, centralized_six_robots,
                      ten_robots_collision_avoidance, third_scenario_obstacles, six_robots_eight_obstacles,
                      two_robots_no_collision_rows, script_preset, script_names)
from .solver import NmpcSolver, PipelinedSolver, nlpsol, shift, shift_states, cold_start, odometry_to_global, split_swarm, merge_swarm
from .distributed import shard_range, gather_results
from .closed_loop import simulate_closed_loop, simulate_closed_loop_fleets, EpisodeResult, simulate_lidar_closed_loop, LidarEpisodeResult
from .lidar import LidarProblemConfig, LidarSolver, lidar_nlpsol, lidar_v4, lidar_v3, lidar_cold_start, lidar_params
from . import _lib

__all__ = ["ProblemConfig", "NmpcSolver", "PipelinedSolver", "nlpsol", "shift", "shift_states", "cold_start", "odometry_to_global", "split_swarm", "merge_swarm", "shard_range", "gather_results", "simulate_closed_loop", "simulate_closed_loop_fleets", "EpisodeResult", "simulate_lidar_closed_loop", "LidarEpisodeResult",
           "centralized_one_robot", "centralized_two_robots", "centralized_six_robots", "ten_robots_collision_avoidance",
           "third_scenario_obstacles", "six_robots_eight_obstacles", "two_robots_no_collision_rows", "script_preset", "script_names",
           "LidarProblemConfig", "LidarSolver", "lidar_nlpsol", "lidar_v4", "lidar_v3", "lidar_cold_start", "lidar_params"]
