"""Global constants of the reference (configuration.py:1-34), same names and values.  Shapes that the reference
freezes here (BATCH_SIZE, IMG_SIZE) are only defaults in this build: the engine takes them at construction."""
import os
from math import ceil

SEED = 47                                                        # configuration.py:4
DATA_FOLDERS = [os.sep.join(["datasets", "rpg-maker-xp"])]      # :6
DIRECTIONS = ["back", "left", "front", "right"]                 # :8
DIRECTION_BACK, DIRECTION_LEFT, DIRECTION_FRONT, DIRECTION_RIGHT = range(4)   # :9-12
DIRECTION_FOLDERS = [f"{i}-{name}" for i, name in enumerate(DIRECTIONS)]     # :13
DATASET_SIZES = [294]                                            # :15
DATASET_SIZE = sum(DATASET_SIZES)
TRAIN_PERCENTAGE = 0.85                                          # :17
TRAIN_SIZES = [ceil(n * TRAIN_PERCENTAGE) for n in DATASET_SIZES]
TRAIN_SIZE = sum(TRAIN_SIZES)
TEST_SIZES = [DATASET_SIZES[i] - TRAIN_SIZES[i] for i in range(len(DATASET_SIZES))]
TEST_SIZE = sum(TEST_SIZES)
BUFFER_SIZE = DATASET_SIZE
BATCH_SIZE = 4                                                   # :24
IMG_SIZE = 64                                                    # :26
INPUT_CHANNELS = 4
OUTPUT_CHANNELS = 4
MAX_PALETTE_SIZE = 256                                           # :31
INVALID_INDEX_COLOR = [255, 0, 220, 255]                         # :32
TEMP_FOLDER = "temp-side2side"                                   # :34
