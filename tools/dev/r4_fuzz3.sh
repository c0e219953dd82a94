# round-4 end: sampler / block / forward-only U-Net sweeps on the final sources (the sampling path's share of the round's changes:
# LinearAttention context / output kernels, F(2x2) tile choice without a norm, small NT GEMM of the time MLP)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4_fuzz_sweeps3.txt
echo "# tools/fuzz_sampler.py (seed 5), tools/fuzz_block.py (seed 6 x 150), tools/fuzz_unet.py forward-only --big (seed 21 x 12), tools/fuzz_vae.py (seed 3)" > $out
python3 tools/fuzz_sampler.py --seed 5 2>&1 | grep -v amdgpu.ids | grep -v "^OK" | tail -3 >> $out; echo sampler done
python3 tools/fuzz_block.py --seed 6 --n 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $out; echo block done
python3 tools/fuzz_unet.py --big --seed 21 --n 12 2>&1 | grep -v amdgpu.ids | grep -v "^OK" | tail -3 >> $out; echo unet done
python3 tools/fuzz_vae.py --seed 3 2>&1 | grep -v amdgpu.ids | grep -v "^OK" | tail -3 >> $out
cat $out
