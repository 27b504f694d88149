from .likelihoods import ExpectedCost, ExponentiatedUtility, GaussianLikelihood  # noqa: F401
from .mpf import MPF  # noqa: F401
from .svgd import get_gmm  # noqa: F401
from .svmpc import SVMPC  # noqa: F401
