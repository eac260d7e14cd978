#!/bin/bash
# round 4: what do the waves of attention_pk_kernel<9> wait for?  two PMC passes over tools/microbench.py attn
bash tools/pmc.sh r04q_attn attention_pk attn > gpurun_out/r04q_attn_pmc.txt 2>&1
cat gpurun_out/r04q_attn_pmc.txt | cut -c1-900
