from .trainor import Trainor  # noqa: F401
from .utils import create_data_loader, create_model  # noqa: F401
from .validator import Validator  # noqa: F401

# One process per GPU is this build's only execution model: ``Trainor`` / ``Validator`` already shard the data, all-reduce the
# flat gradient over RCCL and gather validation outputs when WORLD_SIZE > 1, so the reference's accelerate-based variants
# (executors/trainor_accelerate.py:107, validator_accelerate.py) are the same classes under their reference names.
TrainorAccelerate = Trainor
ValidatorAccelerate = Validator
