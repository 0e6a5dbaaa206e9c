set -e
timeout -k 10 600 python -m pytest tests/test_resdet.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -4
bash tools/probe/ab_gn.sh
