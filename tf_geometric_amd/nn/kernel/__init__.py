# coding=utf-8
from .map_reduce import (aggregate_neighbors, identity_mapper, neighbor_count_mapper, gcn_mapper, sum_reducer,
                         mean_reducer, max_reducer, sum_updater, identity_updater)
from .segment import (segment_softmax, segment_count, segment_op_with_pad, segment_sum, segment_mean, segment_max,
                      segment_min)
