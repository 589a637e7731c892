class ScatterDensityArtist: pass
