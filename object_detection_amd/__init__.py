"""MI355X-native detection box-ops (target assignment, threshold/top-k/decode, per-class NMS)
behind the call signatures of rs1004/object_detection's models and predictor."""
from . import anchors, ops  # noqa: F401
from .models import FCOS, MODELS, SSD300, SSD512, RetinaNet, YoloV3  # noqa: F401
from .ops import batched_nms, box_convert, box_iou  # noqa: F401
from .predictor import Predictor  # noqa: F401
from .binding import bind_boxops  # noqa: F401
from .pipeline import StreamRing  # noqa: F401
