"""Live-camera ingest in front of the hot path (the part that needs no ROS).

The reference's production mode (load_data.py:2433-2444) takes a sensor_msgs/PointCloud2 from the RealSense d435i,
keeps every 4th point starting at index 1, rotates camera axes into lidar axes and lifts the cloud by 1 m:

    r  = R.from_euler('y', -90, degrees=True).as_dcm()      # scipy Rotation (as_dcm is today's as_matrix)
    r2 = R.from_euler('x',  90, degrees=True).as_dcm()
    points = np.dot(np.dot(points, r), r2) + [0.0, 0.0, 1.0]

`ros_numpy` / `rospy` are not available here: `pointcloud2_to_xyz` restates what
`ros_numpy.point_cloud2.pointcloud2_to_xyz_array` does with a sensor_msgs/PointCloud2 (third-party dependency of the
reference, not vendored: eric-wieser/ros_numpy, `point_cloud2.py` -- fields -> structured dtype with the message's
offsets, one record per `point_step` bytes, rows of `row_step` bytes, NaN points dropped, x y z stacked) on the
message's plain attributes, so the rest of this module starts from the [N,3] xyz array either way.  `realsense_to_lidar64` evaluates the reference's expression itself --
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

# sensor_msgs/PointField datatype codes -> numpy (INT8 1 ... FLOAT64 8)
_PF_TYPES = {1: "i1", 2: "u1", 3: "i2", 4: "u2", 5: "i4", 6: "u4", 7: "f4", 8: "f8"}


def pointcloud2_to_xyz(data, width, height, point_step, row_step, fields, is_bigendian=False, remove_nans=True):
    """sensor_msgs/PointCloud2 -> [N,3] array of its x y z fields (their own dtype, float32 for the d435i), what
    `ros_numpy.point_cloud2.pointcloud2_to_xyz_array(msg)` returns (load_data.py:2433).

    data: the message's byte buffer; fields: iterable of (name, offset, datatype, count) -- `(f.name, f.offset,
    f.datatype, f.count)` of `msg.fields`.  Points with a non-finite coordinate are dropped (remove_nans), in
    message order (row-major over height x width), as ros_numpy does."""
    fl = sorted(((str(n), int(o), int(t), int(c)) for n, o, t, c in fields), key=lambda f: f[1])
    names = {f[0] for f in fl}
    if not {"x", "y", "z"} <= names:
        raise ValueError(f"PointCloud2 without x/y/z fields: {sorted(names)}")
    order = ">" if is_bigendian else "<"
    spec = {"names": [], "formats": [], "offsets": [], "itemsize": int(point_step)}
    for name, off, typ, cnt in fl:
        if typ not in _PF_TYPES:
            raise ValueError(f"PointField {name}: unknown datatype {typ}")
        base = np.dtype(order + _PF_TYPES[typ])
        if off + base.itemsize * max(cnt, 1) > point_step:
            raise ValueError(f"PointField {name} (offset {off}) does not fit point_step {point_step}")
        spec["names"].append(name)
        spec["formats"].append(base if cnt <= 1 else (base, (cnt,)))
        spec["offsets"].append(off)
    dt = np.dtype(spec)
    width, height, row_step = int(width), int(height), int(row_step)
    if row_step < width * point_step:
        raise ValueError(f"row_step {row_step} < width {width} x point_step {point_step}")
    buf = np.frombuffer(data, dtype=np.uint8)
    if buf.size < height * row_step:
        raise ValueError(f"PointCloud2 data holds {buf.size} bytes, {height} rows of {row_step} needed")
    rows = buf[:height * row_step].reshape(height, row_step)[:, :width * point_step]
    rec = np.ascontiguousarray(rows).reshape(-1).view(dt)            # height * width records, row-major
    xyz = np.stack([rec["x"], rec["y"], rec["z"]], axis=-1)
    if remove_nans:
        xyz = xyz[np.isfinite(xyz).all(axis=1)]
    return xyz


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
