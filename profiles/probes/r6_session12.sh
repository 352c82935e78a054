#!/bin/bash
export TMPDIR=/tmp
python3 profiles/probes/f4_k1_size_probe.py 2>&1 | grep -v amdgpu.ids
