# sourced by the tools/exp_*.sh experiments: they patch kernel sources / parameters in place and rebuild
# libhipdeflate.so, so whatever they touch is put back (and the library rebuilt from the real sources)
# when the script ends, however it ends.  usage: exp_guard <file> ...
exp_guard() {
	_EXP_BAK=$(mktemp -d)
	_EXP_FILES="$*"
	for f in $_EXP_FILES; do mkdir -p "$_EXP_BAK/$(dirname "$f")" && cp -p "$f" "$_EXP_BAK/$f"; done
	trap 'for f in $_EXP_FILES; do cp -p "$_EXP_BAK/$f" "$f"; done; rm -rf "$_EXP_BAK"; make -s -C 7bgzf_amd/csrc > /dev/null 2>&1 || true' EXIT
}
