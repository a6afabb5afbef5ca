#!/bin/bash
# handles x copy threads for tools/host_pipeline.cpp at B = 4096 (and 256)
bash scripts/gpu_h2h_cpp.sh > /dev/null 2>&1
for B in 4096 256; do
for mode in pageable pinned; do for h in 3 4 6 8; do for th in 8 16 32; do
  [ $mode = pinned ] && [ $th != 8 ] && continue
  nb=$((300000 / B))
  echo -n "B=$B $mode handles=$h threads=$th: "; /tmp/host_pipeline /tmp/frames.bin $B $nb $h $mode - $th | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g frames/s  %.1f GB/s' % (d['frames_per_s'], d['upload_GBs']))"
done; done; done; done
