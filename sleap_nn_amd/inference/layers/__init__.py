from sleap_nn_amd.inference.layers.base import InferenceLayer  # noqa: F401
from sleap_nn_amd.inference.layers.bottomup import BottomUpLayer  # noqa: F401
from sleap_nn_amd.inference.layers.bottomup_multiclass import BottomUpMultiClassLayer  # noqa: F401
from sleap_nn_amd.inference.layers.centered_instance import CenteredInstanceLayer  # noqa: F401
from sleap_nn_amd.inference.layers.centroid import CentroidLayer  # noqa: F401
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig  # noqa: F401
from sleap_nn_amd.inference.layers.single_instance import SingleInstanceLayer  # noqa: F401
from sleap_nn_amd.inference.layers.topdown import TopDownLayer  # noqa: F401
from sleap_nn_amd.inference.layers.topdown_multiclass import CenteredInstanceMultiClassLayer, TopDownMultiClassLayer  # noqa: F401
