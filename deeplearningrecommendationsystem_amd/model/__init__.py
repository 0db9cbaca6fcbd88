"""Drop-in counterparts of the reference's ``model/*.py`` classes: same class
names, constructor arguments, parameter names (state_dict keys) and ``forward``
signatures; forward/backward run on libctrhip's gfx950 kernels."""
from .mf import MatrixFactorization
from .neuralcf import NeuralCF
from .deepfm import DeepFM
from .pnn import PNN
from .ffm import FFM
from .deepcrossing import DeepCrossing
from .din import DIN
from .dien import DIEN
from .embedding_stage import EmbeddingStage
from .deepcross import DeepCross
from .widedeep import WideDeep
from .lr import LogisticRegression
from .nfm import NFM
from .afm import AFM
from .autorec import AutoRec

__all__ = ["MatrixFactorization", "NeuralCF", "DeepFM", "PNN", "FFM", "DeepCrossing", "DIN", "DIEN", "EmbeddingStage", "DeepCross", "WideDeep", "LogisticRegression", "NFM", "AFM", "AutoRec"]
