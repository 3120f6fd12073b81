#!/bin/bash
# A longer parity hunt than the default suite (GPU box): more seeds of the ALS sweeps (once with the drawn arithmetic, once with
# the three-term f16 split forced where it is built: 49..64 features), of the top-N sweep, of the text-ingest corpora (one sort
# pipeline and user-id range by user-id range), against their oracles.  Output: gpurun_out/r6_parity_evidence.txt
export MALS_FUZZ_SEEDS=${MALS_FUZZ_SEEDS:-3000} MALS_TOPN_SEEDS=${MALS_TOPN_SEEDS:-1000} MALS_TEXT_SEEDS=${MALS_TEXT_SEEDS:-400}
F="^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL\|NCCL\|amdgpu.ids\|^$"
{
  echo "# MALS_FUZZ_SEEDS=$MALS_FUZZ_SEEDS MALS_TOPN_SEEDS=$MALS_TOPN_SEEDS MALS_TEXT_SEEDS=$MALS_TEXT_SEEDS (one MI355X, the library of this commit)"
  echo "## ALS sweeps (tests/test_gpu_fuzz.py)"
  python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "$F" | grep "parity sweeps\|passed\|failed\|FAILED" | tail -5
  echo "## ALS sweeps with MALS_FUZZ_GRAMIAN_MODE=3 (three f16 terms per operand wherever the case has 49..64 features)"
  MALS_FUZZ_GRAMIAN_MODE=3 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider -k "seeded_configuration_sweep" 2>&1 | grep -v "$F" | grep "parity sweeps\|passed\|failed\|FAILED" | tail -5
  echo "## top-N sweep (tests/test_gpu_topn.py::test_seeded_recommend_sweep, exact ranking and score bits vs oracle/topn_oracle.py)"
  python -m pytest tests/test_gpu_topn.py -m gpu -q -k "seeded_recommend_sweep" -p no:cacheprovider 2>&1 | grep "passed\|failed\|FAILED" | tail -5
  echo "## text ingest corpora (tests/test_gpu_ingest_text.py::test_fuzzed_corpus_matches_oracle, bit-exact vs oracle/ingest_text_oracle.py)"
  python -m pytest tests/test_gpu_ingest_text.py -m gpu -q -k "fuzzed_corpus" -p no:cacheprovider 2>&1 | grep "passed\|failed\|FAILED" | tail -5
  echo "## the same through the partitioned finish (tests/test_gpu_ingest_big.py::test_fuzzed_text_corpus_in_ranges_matches_oracle)"
  python -m pytest tests/test_gpu_ingest_big.py -m gpu -q -k "fuzzed_text_corpus" -p no:cacheprovider 2>&1 | grep "passed\|failed\|FAILED" | tail -5
} > gpurun_out/r6_parity_evidence.txt 2>&1
cat gpurun_out/r6_parity_evidence.txt
