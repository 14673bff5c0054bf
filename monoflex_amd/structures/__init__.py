from .image_list import ImageList, to_image_list
from .params_3d import Calibration, ParamsList
