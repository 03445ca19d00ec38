# coding=utf-8
from .common_pool import mean_pool, sum_pool, max_pool, min_pool
from .topk_pool import topk_pool
