from .collate import custom_collate
from .stack_dataset import StackDataset, synthetic_exposure_stack
