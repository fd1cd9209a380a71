from .collate import custom_collate
from .stack_dataset import ArtefactStack, StackDataset, synthetic_exposure_stack
