#!/usr/bin/env python
"""Training-step timing of the native decoder (BASELINE config 5 shape, one GPU): bench.py's `train_step` leg on its
own (compute_loss forward + backward + AdamW on B utterances x T frames, per-item t, dropout on).  Prints one JSON line
(not the headline metric)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--fused-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of train.py:60's default construction")
    args = ap.parse_args()
    import oracle
    from bench import train_step_leg
    print(json.dumps(train_step_leg(torch.device("cuda", 0), oracle.make_state_dict(1234), args.batch, args.frames, args.dtype, args.steps,
                                    dropout=not args.no_dropout, fused_adamw=args.fused_adamw)))


if __name__ == "__main__":
    main()
