cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h; O=gpurun_out/r3h
python bench.py --campaign falcon9 > $O/campaign_falcon9.json 2> $O/campaign.err
timeout 600 python -m pytest tests/test_gpu_falcon9_closed_loop.py tests/test_gpu_falcon9.py -q -x -k "f32 or fast or campaign or config5" -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
python tools/falcon9_k1.py > $O/falcon9_k1.txt 2>&1
cat $O/campaign_falcon9.json; grep -v "^$" $O/pytest.log | tail -15; tail -5 $O/falcon9_k1.txt
