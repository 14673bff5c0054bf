"""KITTI text formats (reference data/datasets/kitti_utils.py:61-97 Object3d, :160-218 Calibration, :443-447 read_label).

Only parsing happens on the host: a label line becomes one row of 14 float64 values, exactly the numbers the reference's
Object3d holds before any arithmetic; every derived quantity (alpha, corners, projections, ...) is computed by
mfx_kitti_encode_targets on the device."""
import numpy as np

from ...structures.params_3d import Calibration as _Calibration

TYPE_ID_CONVERSION = {"Car": 0, "Pedestrian": 1, "Cyclist": 2, "Van": -4, "Truck": -4, "Person_sitting": -2,
                      "Tram": -99, "Misc": -99, "DontCare": -1}                       # config/__init__.py:3-13
RECORD_WIDTH = 14   # cls_id, truncation, occlusion, xmin, ymin, xmax, ymax, h, w, l, tx, ty, tz, ry


def read_calib_file(path):
    """{key: float array} for every `key: v v v ...` line; non-numeric lines are ignored (kitti_utils.py:197-214)."""
    data = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip()
            if not line:
                continue
            key, value = line.split(":", 1)
            try:
                data[key] = np.array([float(x) for x in value.split()])
            except ValueError:
                pass
    return data


class Calibration(_Calibration):
    """Camera matrix of the left (P2) or right (P3) colour camera, from a KITTI calib file."""

    def __init__(self, calib_filepath, use_right_cam=False):
        calibs = read_calib_file(calib_filepath)
        super().__init__(np.reshape(calibs["P3"] if use_right_cam else calibs["P2"], [3, 4]))

    @classmethod
    def from_matrix(cls, P):
        self = cls.__new__(cls)
        _Calibration.__init__(self, P)
        return self

    def flipped(self, img_w):
        """Calibration after the horizontal flip (augmentations.py:70-75)."""
        P = self.P.copy()
        P[0, 2] = img_w - P[0, 2] - 1
        P[0, 3] = -P[0, 3]
        return Calibration.from_matrix(P)


def parse_label_line(line):
    """One label_2 line -> (type name, 14-value record with cls_id = TYPE_ID_CONVERSION[type])."""
    data = line.split(" ")
    if len(data) < 15:
        raise ValueError("KITTI label line needs 15 fields, got %d: %r" % (len(data), line))
    v = [float(x) for x in data[1:15]]
    if data[0] not in TYPE_ID_CONVERSION:
        raise KeyError("unknown KITTI object type %r" % data[0])
    rec = [float(TYPE_ID_CONVERSION[data[0]]), v[0], float(int(v[1])), v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13]]
    return data[0], rec


def read_label_records(path_or_lines, classes):
    """read_label + filtrate_objects (kitti_utils.py:443-447, kitti.py:202-216): records of the objects whose type is in
    `classes`, in file order, as an (n, 14) float64 array."""
    if isinstance(path_or_lines, str):
        with open(path_or_lines) as f:
            lines = [l.rstrip() for l in f]
    else:
        lines = [l.rstrip() for l in path_or_lines]
    rows = []
    for line in lines:
        if not line:
            continue
        typ, rec = parse_label_line(line)
        if typ in classes:
            rows.append(rec)
    return np.asarray(rows, dtype=np.float64).reshape(-1, RECORD_WIDTH)
