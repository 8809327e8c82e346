#!/bin/bash
mkdir -p gpurun_out/r05c
timeout -k 10 500 python tools/fm_stamps.py > gpurun_out/r05c/fm_stamps.txt 2>&1; rc=$?
tail -40 gpurun_out/r05c/fm_stamps.txt | cut -c1-250; exit $rc
