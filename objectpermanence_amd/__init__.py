"""objectpermanence_amd - MI355X-native OPNet reasoner hot path of ofrikleinfeld/ObjectPermanence."""
from .learned_models import (AbstractCaterModel, BaselineLstm, NonLinearLstm, OPNet, OPNetLstmMlp,  # noqa: F401
                             TransformerLstm)
from .models_factory import ModelsFactory  # noqa: F401
from .optim import FusedAdam, l1_mean  # noqa: F401
