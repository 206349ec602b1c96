#!/bin/bash
# Round 5, spare GPU minutes: more seeds of the three fuzzers on the final device sources
cd $GRAFT_REPO_ROOT
python -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash())"
timeout 900 python tools/fuzz_parity.py 7150 250
timeout 400 python tools/fuzz_sampler.py 17080 320
timeout 500 python tools/fuzz_lean.py 13060 60
timeout 500 python tools/fuzz_queues.py 11060 40
