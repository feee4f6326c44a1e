"""Live-camera ingest in front of the hot path (the part that needs no ROS).

The reference's production mode (load_data.py:2433-2444) takes a sensor_msgs/PointCloud2 from the
RealSense d435i, keeps every 4th point starting at index 1, rotates camera axes into lidar axes
(scipy Rotation: -90 deg about y, then +90 deg about x, applied as row-vector products) and lifts the
cloud by 1 m.  `ros_numpy` / `rospy` are not available here, so this module starts from the [N,3] xyz
array `pointcloud2_to_xyz_array` would return; the two rotations compose to a signed axis
permutation, applied exactly (the reference's matrices carry cos(90 deg) = 6.1e-17 instead of 0).
"""
import numpy as np

# R_y(-90 deg) @ R_x(+90 deg), exact entries; p_lidar = p_camera @ CAMERA_TO_LIDAR
_RY = np.array([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]])
_RX = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
CAMERA_TO_LIDAR = _RY @ _RX
SENSOR_HEIGHT = 1.0


def realsense_to_lidar(points_xyz, decimate=4, first=1, lift=SENSOR_HEIGHT, dtype=np.float32):
    """[N,3] camera-frame points -> [ceil((N-first)/decimate),3] lidar-frame points (x depth, y left, z up),
    ready for `Engine.detect` / `points_to_voxel` (float32 like every cloud the hot path takes)."""
    p = np.asarray(points_xyz)
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError(f"expected an [N,3] xyz array, got {p.shape}")
    p = p[first::decimate].astype(np.float64)
    out = np.empty_like(p)
    for j in range(3):   # signed permutation: one source column per output column
        src = int(np.argmax(np.abs(CAMERA_TO_LIDAR[:, j])))
        out[:, j] = p[:, src] * CAMERA_TO_LIDAR[src, j]
    out[:, 2] += lift
    return out.astype(dtype)
