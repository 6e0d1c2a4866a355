#!/bin/bash
# closing session of a round on the final code: the whole GPU suite, smoke(), and the randomised differential sweeps
# (GPU Huffman reader and both file-level pipelines against the host reader, `oyuv convert` against the restated Oconv), each
# into gpurun_out/${TAG}_*.
set -e
TAG=${1:-r04z}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -40 gpurun_out/${TAG}_pytest.log; exit 1; }
tail -3 gpurun_out/${TAG}_pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1 || { tail -20 gpurun_out/${TAG}_smoke.log; exit 1; }
tail -1 gpurun_out/${TAG}_smoke.log
{
echo "== tools/stress_hdec.py --cases 400 --mutations 800 --seed 11"
timeout -k 10 900 python tools/stress_hdec.py --cases 400 --mutations 800 --seed 11
echo "== tools/stress_pipeline.py --cases 300 --seed 12"
timeout -k 10 900 python tools/stress_pipeline.py --cases 300 --seed 12
echo "== tools/stress_convert.py --cases 3000 --seed 13"
timeout -k 10 600 python tools/stress_convert.py --cases 3000 --seed 13
} 2>&1 | tee gpurun_out/${TAG}_stress.txt
