"""dev: the cfg3b gather leg of bench.py under the library named by CTRHIP_LIB (A/B of build variants)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.gather_stage_leg(torch.device("cuda:0"), reps=40)
print(os.environ.get("CTRHIP_LIB", "default").split("_")[-1], "uniform fwd %.1f us (%.3f)  zipf fwd %.1f us (%.3f)  bwd %.1f / %.1f us" % (
    r["avg_us"], r["frac"], r["zipf"]["avg_us"], r["zipf"]["frac"], r["scatter_bwd"]["uniform"]["avg_us"], r["scatter_bwd"]["zipf"]["avg_us"]))
