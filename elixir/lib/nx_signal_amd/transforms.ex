defmodule NxSignalAMD.Transforms do
  @moduledoc """
  `NxSignal.Transforms.fft_nd/2` and `ifft_nd/2` (lib/nx_signal/transforms.ex:5-21): `Nx.fft` / `Nx.ifft` folded over
  `axes:` with `lengths:` — the whole fold runs in HBM (tiled transposes between axes, four-step rows beyond 8192 points,
  Bluestein rows for non-power-of-two lengths).
  """
  alias NxSignalAMD.NIF

  def fft_nd(tensor, opts \\ []), do: run(tensor, opts, 0)
  def ifft_nd(tensor, opts \\ []), do: run(tensor, opts, 1)

  defp run(%Nx.Tensor{} = tensor, opts, inverse) do
    opts = Keyword.validate!(opts, axes: [-1], lengths: nil)
    axes = opts[:axes]
    shape = Nx.shape(tensor) |> Tuple.to_list()
    rank = length(shape)
    axes_n = Enum.map(axes, fn ax -> if ax < 0, do: ax + rank, else: ax end)
    lengths = (opts[:lengths] || List.duplicate(nil, length(axes))) |> Enum.zip(axes_n) |> Enum.map(fn {l, ax} -> l || Enum.at(shape, ax) end)

    {bin, is_real} =
      case Nx.type(tensor) do
        {:c, 64} -> {Nx.to_binary(tensor), 0}
        {:c, _} -> raise ArgumentError, "only c64 complex tensors are supported"
        {:f, 64} -> raise ArgumentError, "f64 tensors are not supported by the MI355X path"
        _ -> {tensor |> Nx.as_type(:f32) |> Nx.to_binary(), 1}
      end

    {:ok, out} = NIF.fft_nd(NxSignalAMD.context(), bin, is_real, shape, axes_n, lengths, inverse) |> NxSignalAMD.unwrap!()

    out_shape =
      Enum.zip(axes_n, lengths) |> Enum.reduce(shape, fn {ax, l}, acc -> List.replace_at(acc, ax, l) end) |> List.to_tuple()

    Nx.from_binary(out, :c64) |> Nx.reshape(out_shape)
  end
end
