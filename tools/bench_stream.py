#!/usr/bin/env python3
"""Streaming (chunked decoder) latency on one MI355X -- thin wrapper around `bench.py --stream`
(the benchmark and its CPU-port comparison leg live in bench.py, the one place outside tests/ that
may time the CPU restatement).  Usage: tools/bench_stream.py [--model M] [--phonemes N] [--chunk C]
[--pad P] [--cpu] [--unfused]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAP = {"--phonemes": "--stream-phonemes", "--chunk": "--stream-chunk", "--pad": "--stream-pad",
       "--reps": "--stream-reps", "--cpu": "--stream-cpu", "--unfused": "--stream-unfused"}
argv = [MAP.get(a, a) for a in sys.argv[1:]]
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--stream"] + argv))
