#!/usr/bin/env python3
"""Floors of whisper_stft_mel_kernel by exclusion: the shipped kernel next to builds of the same source with phases left
out (STTFE_FLOOR in csrc/whisper_logmel.hip; `make -C speech_to_text_amd/csrc floors` builds them here), on the headline
workload (1024 x 30 s clips, 80 mel bins).  Same lane layout, same tables in LDS, same two workgroups per CU:
    1  the transform alone (window, 20 x 20 Good-Thomas FFT with the slab exchange, A/B separation, |.|^2):
       no PCM loads, no power panel, no barriers, no mel projection, no stores
    2  PCM loads + transform
    3  everything but the transform (loads, panel, barriers, banded MFMA mel projection, log epilogue, stores)
    4  the whole kernel, the 45 MFMAs per wave and tile replaced by one packed multiply-add per operand pair (wrong features)
    5  the whole kernel without the two workgroup barriers per tile (wrong features);  6  both
    7  the whole kernel without the feature stores;  8  without the mel phase (barriers and panel kept)
    9  the mel units' operands from registers instead of LDS;  10  no logarithm, no min / max keys;  11  7 + 9 + no logarithm
Usage: python3 tools/ubench/whisper_floor.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = [os.path.join(ROOT, "speech_to_text_amd", "libsttfe.so")] + [os.path.join(ROOT, "tools", "ubench", f"whisper_floor_{i}.so") for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)]
env = dict(os.environ, AB_TPB=os.environ.get("AB_TPB", "0"))
sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_bench.py")] + libs, env=env).returncode)
