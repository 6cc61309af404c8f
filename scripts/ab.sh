SADVIO_LM=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_replay.py tests/test_gpu_vio.py -m gpu -q -x 2>&1 | grep -v "RCCL\|NCCL" | tail -3
echo "== LM"; timeout 300 python scripts/gpu_time.py 64 2>&1 | grep -v "RCCL\|NCCL" | head -5
