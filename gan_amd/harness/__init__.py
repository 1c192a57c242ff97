"""Minimal restatement of the reference's model graphs and step order on top of gan_amd.nn, used to drive and
measure the convolution kernels (bench.py, tests).  Data loading, logging and checkpointing of the reference
scripts are out of scope."""
