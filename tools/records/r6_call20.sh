#!/bin/bash
# Round 6, GPU call 20: tools/decode_fuzz_long.py again (call 19 stopped at a refused input: one-channel heads in channels_last; the
# bindings now let such a tensor follow its partner), 2500 cases; the fuzz / binding / parity suites on the changed bindings.
O=gpurun_out/r6c20; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python tools/decode_fuzz_long.py --seeds 0:2500 ) > $O/decode_fuzz_long.txt 2>&1; grep -v amdgpu.ids $O/decode_fuzz_long.txt | tail -12 | cut -c1-400
( time timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_compiled_binding.py tests/test_gpu_parity.py tests/test_gpu_loss.py -m gpu -q ) > $O/pytest.txt 2>&1; grep -v amdgpu $O/pytest.txt | tail -4
