cd $GRAFT_REPO_ROOT
for m in fp32 bf16x6; do for k in "" mrd mpd mel "mpd,mrd" "mpd,mrd,mel"; do MODE=$m KO=$k python tools/knockout.py 2>&1 | tail -1; done; done
