from .base_model import BaseModel  # noqa: F401
from .base_trainer import BaseTrainer  # noqa: F401
from .base_dataloader import BaseDataLoader, DataPrefetcher  # noqa: F401
from .base_dataset import BaseDataSet  # noqa: F401
