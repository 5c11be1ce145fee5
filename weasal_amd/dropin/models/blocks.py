from weasal_amd.blocks import *  # noqa: F401,F403
from weasal_amd.blocks import (KPConv, BatchNormBlock, UnaryBlock, SimpleBlock, SimpleBlock2,  # noqa: F401
                               ResnetBottleneckBlock, GlobalAverageBlock, NearestUpsampleBlock, MaxPoolBlock,
                               block_decider, gather, radius_gaussian, closest_pool, max_pool, global_average,
                               spatial_att, channel_att, multi_path_att, global_average_block, ele_att)
import torch  # noqa: F401  (the reference's architectures.py relies on names star-imported from blocks)
import torch.nn as nn  # noqa: F401
