from .util import *  # noqa: F401,F403
from .diff_op import *  # noqa: F401,F403
from .functions import *  # noqa: F401,F403
from .registration import *  # noqa: F401,F403
from .sampler import *  # noqa: F401,F403
from .transformation import *  # noqa: F401,F403
