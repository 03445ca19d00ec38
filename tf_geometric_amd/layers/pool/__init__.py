# coding=utf-8
from .readout import CommonPool, MeanPool, SumPool, MaxPool, MinPool
