from .pandas import (  # noqa: F401
    df_to_temporal_graph,
    read_csv_path_data,
    read_csv_temporal_graph,
    temporal_graph_to_df,
    write_csv,
)
