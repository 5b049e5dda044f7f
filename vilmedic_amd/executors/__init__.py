from .trainor import Trainor  # noqa: F401
from .utils import create_data_loader, create_model  # noqa: F401
from .validator import Validator  # noqa: F401
