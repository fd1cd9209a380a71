"""the real-kernel path of bench.c5_strong_block on one GPU (world = 1): the code the driver's --gpus N line runs per rank"""
import argparse, json, sys, os, torch
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda:0")
args = argparse.Namespace(exposures=32, global_size=4096, steps=10, warmup=3)
print(json.dumps(bench.c5_strong_block(args, 0, 1, dev)))
