"""bocf_amd: MI355X-native GP-posterior + composite-acquisition hot path of BOCF behind the
reference's multi_outputGP / AcquisitionBase plug-in surface.  See DESIGN.md."""
from . import _ffi, kern  # noqa: F401
from .acquisition_optimizer import AcquisitionOptimizer, Design_space  # noqa: F401
from .acquisitions import EI, PI, AcquisitionBase, maEI, maPI, uEI_noiseless, uPI  # noqa: F401
from .multi_outputGP import multi_outputGP  # noqa: F401
from .utility import ExpectationUtility, ParameterDistribution, Utility  # noqa: F401

__version__ = "0.1.0"
