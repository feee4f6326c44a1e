"""Live-camera ingest in front of the hot path (the part that needs no ROS).

The reference's production mode (load_data.py:2433-2444) takes a sensor_msgs/PointCloud2 from the RealSense d435i,
keeps every 4th point starting at index 1, rotates camera axes into lidar axes and lifts the cloud by 1 m:

    r  = R.from_euler('y', -90, degrees=True).as_dcm()      # scipy Rotation (as_dcm is today's as_matrix)
    r2 = R.from_euler('x',  90, degrees=True).as_dcm()
    points = np.dot(np.dot(points, r), r2) + [0.0, 0.0, 1.0]

`ros_numpy` / `rospy` are not available here, so this module starts from the [N,3] xyz array
`pointcloud2_to_xyz_array` would return.  `realsense_to_lidar64` evaluates the reference's expression itself --
same scipy matrices (their cos(90 deg) entries are 6.1e-17, not 0), same two float64 products, same addition --
and is bit-identical to it.  `realsense_to_lidar` hands the engine float32 points (the voxeliser's input type):

  * x_lidar = z_cam and y_lidar = -x_cam are float32 values to begin with (the 6e-17-weighted terms vanish in the
    cast), so their cells are the reference's;
  * z_lidar = -y_cam + 1.0 is rounded to float32, which can move a point across a voxel edge only if the float64
    value lies within half a float32 ulp of it: for the shipped grid (z edges at -3, 1, 5) that is 0 < y_cam < 3e-8
    or y_cam within 1.2e-7 of 4 -- values a depth camera does not produce.  `cells_agree` checks a cloud for it.
"""
import numpy as np

SENSOR_HEIGHT = 1.0


def _matrices():
    from scipy.spatial.transform import Rotation as R
    r = R.from_euler('y', -90, degrees=True)
    r2 = R.from_euler('x', 90, degrees=True)
    as_m = "as_matrix" if hasattr(r, "as_matrix") else "as_dcm"
    return getattr(r, as_m)(), getattr(r2, as_m)()


def realsense_to_lidar64(points_xyz, decimate=4, first=1, lift=SENSOR_HEIGHT):
    """The reference's arithmetic, float64 out (load_data.py:2434-2443)."""
    p = np.asarray(points_xyz)
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError(f"expected an [N,3] xyz array, got {p.shape}")
    r, r2 = _matrices()
    pts = p[first::decimate]
    pts = np.dot(pts, r)
    pts = np.dot(pts, r2)
    return pts + [0.0, 0.0, lift]


def realsense_to_lidar(points_xyz, decimate=4, first=1, lift=SENSOR_HEIGHT, dtype=np.float32):
    """[N,3] camera-frame points -> [ceil((N-first)/decimate),3] lidar-frame points (x depth, y left, z up),
    ready for `Engine.detect` / `points_to_voxel` (float32 like every cloud the hot path takes)."""
    return realsense_to_lidar64(points_xyz, decimate, first, lift).astype(dtype)


def cells_agree(points64, voxel_size, pc_range):
    """True when the float32 cast of `points64` falls into the same voxels as the float64 values (the reference
    voxelises the float64 array): floor((p - min) / size) evaluated as load_data.py:622 does, both ways."""
    vs, lo = np.asarray(voxel_size, np.float64), np.asarray(pc_range, np.float64)[:3]
    c64 = np.floor((np.asarray(points64, np.float64) - lo) / vs)
    c32 = np.floor((np.asarray(points64).astype(np.float32).astype(np.float64) - lo) / vs)
    return bool(np.array_equal(c64, c32))
