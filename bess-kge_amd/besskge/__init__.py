"""besskge - MI355X-native BESS knowledge-graph-embedding hot path (WIP init)."""
