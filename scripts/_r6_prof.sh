H=$(cat .git_head 2>/dev/null || echo unknown)
bash scripts/profile_bench.sh r06prof $H > gpurun_out/r06prof_ls.txt 2>&1
bash scripts/pmc_sq.sh r06sq > /dev/null 2>&1
tail -5 gpurun_out/r06prof_ls.txt; ls gpurun_out/r06prof gpurun_out/r06sq | head -30
