"""hdl_graph_slam_amd — MI355X-native scan-matching backend for hdl_graph_slam (NDT / GICP registration and the
loop-closure candidate batch behind include/hdl_graph_slam/registrations.hpp).  See DESIGN.md / INTEGRATION.md."""
from .registrations import select_registration_method, params_from_rosparams  # noqa: F401
from .registration import RegistrationHIP, DeviceCloud, HgsError, select_best, select_imu_sample  # noqa: F401
from .loop_detector import LoopDetector, KeyFrame, Loop, loop_guess  # noqa: F401
