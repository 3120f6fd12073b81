#!/bin/bash
# A longer parity hunt than the default suite (GPU box): more seeds of the ALS sweeps, of the top-N sweep and of the text-ingest
# corpora, against their oracles.  Output: gpurun_out/r5_parity_evidence.txt
export MALS_FUZZ_SEEDS=${MALS_FUZZ_SEEDS:-1500} MALS_TOPN_SEEDS=${MALS_TOPN_SEEDS:-400} MALS_TEXT_SEEDS=${MALS_TEXT_SEEDS:-150}
{
  echo "# MALS_FUZZ_SEEDS=$MALS_FUZZ_SEEDS MALS_TOPN_SEEDS=$MALS_TOPN_SEEDS MALS_TEXT_SEEDS=$MALS_TEXT_SEEDS (one MI355X, the library of this commit)"
  echo "## ALS sweeps (tests/test_gpu_fuzz.py)"
  python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL\|NCCL\|amdgpu.ids\|^$" | grep "parity sweeps\|passed\|failed\|FAILED" | tail -5
  echo "## top-N sweep (tests/test_gpu_topn.py::test_seeded_recommend_sweep, exact ranking and score bits vs oracle/topn_oracle.py)"
  python -m pytest tests/test_gpu_topn.py -m gpu -q -k "seeded_recommend_sweep" -p no:cacheprovider 2>&1 | grep "passed\|failed\|FAILED" | tail -5
  echo "## text ingest corpora (tests/test_gpu_ingest_text.py::test_fuzzed_corpus_matches_oracle, bit-exact vs oracle/ingest_text_oracle.py)"
  python -m pytest tests/test_gpu_ingest_text.py -m gpu -q -k "fuzzed_corpus" -p no:cacheprovider 2>&1 | grep "passed\|failed\|FAILED" | tail -5
} > gpurun_out/r5_parity_evidence.txt 2>&1
cat gpurun_out/r5_parity_evidence.txt
