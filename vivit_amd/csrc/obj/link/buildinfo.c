const char *vivit_hip_source_hash(void) { return "c333aef5ad12b4e912e11758e1657df8-262e1d34"; }
