python tools/shard_probe.py S300 6 2>&1 | tail -3
python -m pytest tests -m gpu -x -q -k "sharded or parallel or pack" 2>&1 | grep -E " passed| failed|rror" | tail -3
