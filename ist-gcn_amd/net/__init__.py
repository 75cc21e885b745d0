"""Drop-in mirrors of the reference's `net` package (same module names, class names and
constructor signatures): select them in the reference's YAML with `model: istgcn_amd.net.<module>.Model`."""
