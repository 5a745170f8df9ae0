"""Global constants with the reference's names and values (configuration.py:1-34 of fegemo/palette-and-histo-gan).

The notebook and the model classes import these names with `from configuration import *`, so the names are part of the
drop-in surface; the values are the reference's.  Two of them are only DEFAULTS in this build -- BATCH_SIZE and IMG_SIZE
are arguments of the loaders and of the engine (the benchmark configurations use batch 256 and 64/128-pixel sprites).
"""
import math
import os

# -- random seed shared by shuffles, augmentation draws, weight initialisation and dropout streams (:4)
SEED = 47

# -- where the sprite folders live, relative to the working directory (:6); one entry per dataset
DATA_FOLDERS = [os.path.join("datasets", "rpg-maker-xp")]

# -- the four views of a character and their sub-folder names "<index>-<name>" (:8-13)
DIRECTIONS = ["back", "left", "front", "right"]
(DIRECTION_BACK,
 DIRECTION_LEFT,
 DIRECTION_FRONT,
 DIRECTION_RIGHT) = (DIRECTIONS.index(name) for name in DIRECTIONS)
DIRECTION_FOLDERS = ["%d-%s" % pair for pair in enumerate(DIRECTIONS)]

# -- dataset sizes and the 85 % / 15 % train / test split, per dataset and in total (:15-22)
DATASET_SIZES = [294]
TRAIN_PERCENTAGE = 0.85
TRAIN_SIZES = [int(math.ceil(count * TRAIN_PERCENTAGE)) for count in DATASET_SIZES]
TEST_SIZES = [count - train for count, train in zip(DATASET_SIZES, TRAIN_SIZES)]
DATASET_SIZE, TRAIN_SIZE, TEST_SIZE = sum(DATASET_SIZES), sum(TRAIN_SIZES), sum(TEST_SIZES)
BUFFER_SIZE = DATASET_SIZE          # shuffle buffer = the whole set (:23)

# -- batch and sprite geometry (:24-28)
BATCH_SIZE = 4
IMG_SIZE = 64
INPUT_CHANNELS = OUTPUT_CHANNELS = 4          # RGBA in, RGBA out

# -- palette-indexed models (:31-32): palettes are padded to 256 entries with a hot pink no image uses
MAX_PALETTE_SIZE = 256
INVALID_INDEX_COLOR = [255, 0, 220, 255]

# -- folder for logs, checkpoints and generated images (:34)
TEMP_FOLDER = "temp-side2side"


def ceil(x):
    """`from configuration import *` in the notebook relies on `ceil` coming along (experiments.ipynb cell 10)"""
    return int(math.ceil(x))
