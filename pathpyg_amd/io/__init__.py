from .pandas import (  # noqa: F401
    add_edge_attributes,
    add_node_attributes,
    df_to_graph,
    df_to_temporal_graph,
    graph_to_df,
    read_csv_graph,
    read_csv_path_data,
    read_csv_temporal_graph,
    temporal_graph_to_df,
    write_csv,
)
