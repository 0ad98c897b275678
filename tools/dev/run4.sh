cd $GRAFT_REPO_ROOT
AB_DIR=_abx bash tools/dev/run_cfg_ab.sh nopersist persist persist5
